"""Seeded tracklet-graph generators in the data format the reference's callers hand to
`MOTMPNet.forward` (`data.x [N,F] f32`, `data.edge_index [2,E] i64`, `data.edge_attr [E,2] f32`).

They reproduce the *shape* of what inference.py:402-458 / train.py:316-342 build from real
tracklets (the AIC19 data itself is not available offline): cross-camera cartesian edge
lists, column-normalised node embeddings and the [L2 distance, cosine distance] edge
attributes.  Used by tests, bench.py and the golden-vector generator; all pure torch, run
on whatever device the caller asks for.  Recipes: SURVEY.md section 8(d).
"""
from __future__ import annotations

import types
from typing import Sequence

import torch
import torch.nn.functional as F

# tracklets per camera (c006..c009) in the reference's eval/ground_truth_S02.txt
S02_GT_CAMS = (124, 90, 99, 137)
# tracker-scale S02: misc/mtsc_BUPT21 counts for c007/c008, 250 for the two missing cameras
S02_TRACKER_CAMS = (250, 221, 281, 250)


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def camera_edge_index(cam_of_node: torch.Tensor) -> torch.Tensor:
    """Directed cross-camera edges in the reference's order (inference.py:407-413,
    train.py:323-329): for every camera (ascending id), cartesian_prod(nodes in the camera,
    nodes not in it); returned as the same non-contiguous `[E,2].T` view the callers pass."""
    nodes = torch.arange(cam_of_node.numel())
    blocks = []
    for cam in torch.unique(cam_of_node).tolist():
        inside = nodes[cam_of_node == cam]
        outside = nodes[cam_of_node != cam]
        blocks.append(torch.cartesian_prod(inside, outside))
    return torch.cat(blocks, dim=0).T


def appearance_edge_attr(x: torch.Tensor, edge_index: torch.Tensor, chunk: int = 1 << 18) -> torch.Tensor:
    """edge_attr = [pairwise_distance(x[row], x[col]), 1 - cosine_similarity(x[row], x[col])]
    (inference.py:453-456, train.py:340-342), evaluated in chunks so the [E,F] gathers stay small."""
    row, col = edge_index[0], edge_index[1]
    out = torch.empty(row.numel(), 2, dtype=x.dtype, device=x.device)
    for s in range(0, row.numel(), chunk):
        a, b = x[row[s:s + chunk]], x[col[s:s + chunk]]
        out[s:s + chunk, 0] = F.pairwise_distance(a, b)
        out[s:s + chunk, 1] = 1 - F.cosine_similarity(a, b)
    return out


def random_graph(n_nodes: int = 64, n_edges: int = 512, feat: int = 2048, seed: int = 1):
    """BASELINE config 1: randn features, uniformly random (row, col) pairs (unsorted rows,
    duplicates and self-loops possible), rand edge_attr.  Draw order is fixed: x, edge_index, attr."""
    g = _gen(seed)
    x = torch.randn(n_nodes, feat, generator=g)
    edge_index = torch.randint(0, n_nodes, (2, n_edges), generator=g)
    edge_attr = torch.rand(n_edges, 2, generator=g)
    return types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr)


def camera_graph(cams: Sequence[int] = S02_GT_CAMS, feat: int = 2048, seed: int = 2,
                 cam_of_node: torch.Tensor | None = None):
    """BASELINE config 2 stand-in: nodes ordered camera by camera (libs/dataset.py:279-281),
    x = normalize(randn, dim=0) exactly as inference.py:404 does (column-wise!), real edge_attr formula.
    `cam_of_node` overrides the per-camera counts (training-style node order, config 3)."""
    if cam_of_node is None:
        cam_of_node = torch.repeat_interleave(torch.arange(len(cams)), torch.tensor(list(cams)))
    n = cam_of_node.numel()
    x = F.normalize(torch.randn(n, feat, generator=_gen(seed)), p=2, dim=0)
    edge_index = camera_edge_index(cam_of_node)
    return types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=appearance_edge_attr(x, edge_index),
                                 cam_of_node=cam_of_node)


def training_graph(tracklets: Sequence[Sequence[int]], n_ids: int = 100, feat: int = 2048, seed: int = 3):
    """BASELINE config 3 stand-in.  `tracklets` = [(camera, identity), ...] of the training scenes
    (tests/golden/train_tracklets.json).  Samples `n_ids` identities without replacement, one
    node per (identity, camera), nodes ordered by identity then camera (train.py:295-302), so
    `row` is sorted only inside each camera block.  Returns labels for the loss as well."""
    g = _gen(seed)
    ids = sorted({t[1] for t in tracklets})
    pick = [ids[i] for i in torch.randperm(len(ids), generator=g)[:n_ids].tolist()]
    nodes = [(c, i) for i in pick for (c, j) in sorted(tracklets) if j == i]
    cam_of_node = torch.tensor([c for c, _ in nodes])
    id_of_node = torch.tensor([i for _, i in nodes])
    data = camera_graph(feat=feat, seed=seed + 1000, cam_of_node=cam_of_node)
    r, c = data.edge_index[0], data.edge_index[1]
    data.edge_labels = (id_of_node[r] == id_of_node[c]).float()
    data.y = id_of_node
    return data


def stress_graph(n_nodes: int, n_pairs: int, feat: int = 2048, seed: int = 4, device="cpu",
                 with_features: bool = True):
    """BASELINE configs 4/5: `n_pairs` undirected pairs i != j drawn uniformly (duplicates
    allowed), emitted in both directions and sorted by (row, col); randn x, rand edge_attr.
    Generated on `device` (the big ones only make sense on the GPU)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    a = torch.randint(0, n_nodes, (n_pairs,), generator=g, device=device)
    b = torch.randint(0, n_nodes - 1, (n_pairs,), generator=g, device=device)
    b = b + (b >= a).to(b.dtype)                      # uniform over j != i without rejection
    key = torch.cat([a * n_nodes + b, b * n_nodes + a])
    del a, b
    key, _ = torch.sort(key)
    edge_index = torch.stack([key // n_nodes, key % n_nodes])
    del key
    x = torch.randn(n_nodes, feat, generator=g, device=device) if with_features else None
    edge_attr = torch.rand(edge_index.shape[1], 2, generator=g, device=device)
    return types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr)
