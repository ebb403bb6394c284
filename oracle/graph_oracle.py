"""CPU oracle for the callers' graph construction -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, with the same ATen calls, what the reference's loops do between loading the per-tracklet features
and calling the MPN (reference inference.py:402-458, identically train.py:316-342 / :557-561):

    x          = F.normalize(torch.stack(feats), p=2, dim=0)                     inference.py:402-404 (column-wise!)
    edge_index = cat_c cartesian_prod(nodes in camera c, nodes not in c)).T      inference.py:407-413
    edge_label = 1.0 where both ends carry the same identity                     inference.py:446-450
    edge_attr  = [pairwise_distance(x[r], x[c]), 1 - cosine_similarity(x[r], x[c])]   inference.py:453-456

Parity status: PINNED.  tests/golden/make_golden_graph.py drives the reference's own
`inference.inference_precomputed_features` (inference.py:372-458, imported unmodified) with a stub loader and a stub MPN
that captures the `Data` object, asserts that `build()` below reproduces x / edge_index / edge_attr / edge_labels bit
for bit, and stores the fixtures tests/golden/gb_*.npz that tests/test_graph_golden.py checks this file and the HIP
builder against.  train.py:316-342 / :557-561 are the same statements on the same names.  Only tests/ may import it.
"""
import numpy as np
import torch
import torch.nn.functional as F


def build(node_feats: torch.Tensor, cam_ids, node_labels=None, l2norm: bool = True, chunk: int = 1 << 16):
    x = F.normalize(node_feats, p=2, dim=0) if l2norm else node_feats
    cams = np.asarray(cam_ids)
    nodes = np.asarray(range(len(cams)))
    blocks = []
    for c in np.unique(cams):
        inside, outside = nodes[cams == c], nodes[cams != c]
        blocks.append(torch.cartesian_prod(torch.from_numpy(inside), torch.from_numpy(outside)))
    edge_index = torch.cat(blocks, dim=0).T
    row, col = edge_index[0], edge_index[1]
    attr = torch.empty(row.numel(), 2, dtype=x.dtype)
    for s in range(0, row.numel(), chunk):
        a, b = x[row[s:s + chunk]], x[col[s:s + chunk]]
        attr[s:s + chunk, 0] = F.pairwise_distance(a, b)
        attr[s:s + chunk, 1] = 1 - F.cosine_similarity(a, b)
    labels = None
    if node_labels is not None:
        lab = torch.as_tensor(np.asarray(node_labels))
        labels = (lab[row] == lab[col]).float()
    return x, edge_index, attr, labels
