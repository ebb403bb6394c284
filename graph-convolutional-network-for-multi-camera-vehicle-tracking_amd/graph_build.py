"""`build_graph`: the callers' graph construction on the GPU (SURVEY.md 8(f)-1,2).

Replaces what reference inference.py:402-458 (and train.py:316-342) do around the MPN call -- stack + column
normalisation of the 2048-d tracklet features, the cross-camera cartesian edge list, the O(E*N) Python edge-label
loop, and the two 2048-d gathers per edge for `edge_attr` -- by one library call (`mtmc_build_graph`): a Gram matrix
on the matrix cores plus an 8-byte-per-edge epilogue.  Returns a `Data`-like namespace whose attributes feed
`MOTMPNet.forward` directly.
"""
from __future__ import annotations

import types

import numpy as np
import torch

from . import _lib

MAX_NODES = 46000


def camera_tables(cam_ids):
    """Small host-side index tables of the camera structure (O(N * cameras) ints)."""
    cams = np.asarray(cam_ids)
    n = cams.shape[0]
    nodes = np.arange(n, dtype=np.int32)
    uniq = np.unique(cams)
    in_list, out_list, in_off, out_off, block_off = [], [], [0], [0], [0]
    for c in uniq:
        inside, outside = nodes[cams == c], nodes[cams != c]
        in_list.append(inside)
        out_list.append(outside)
        in_off.append(in_off[-1] + inside.size)
        out_off.append(out_off[-1] + outside.size)
        block_off.append(block_off[-1] + inside.size * outside.size)
    return (np.concatenate(in_list).astype(np.int32), np.asarray(in_off, dtype=np.int32),
            np.concatenate(out_list).astype(np.int32) if out_list else np.zeros(0, np.int32),
            np.asarray(out_off, dtype=np.int64), np.asarray(block_off, dtype=np.int64), len(uniq))


def build_graph(node_feats: torch.Tensor, cam_ids, node_labels=None, l2norm: bool = True):
    """node_feats: [N, F] float32 on a ROCm GPU (the stacked per-tracklet ReID features); cam_ids: N camera ids
    (host list / array, as the reference keeps them); node_labels: optional N identities (edge labels for training)."""
    if not (isinstance(node_feats, torch.Tensor) and node_feats.is_cuda):
        raise RuntimeError("mtmc_mpn.build_graph: node_feats must be on a ROCm GPU (no CPU path)")
    if node_feats.dim() != 2 or node_feats.dtype != torch.float32 or node_feats.shape[1] % 32:
        raise RuntimeError("mtmc_mpn.build_graph: node_feats must be float32 [N, F] with F a multiple of 32")
    n, f = node_feats.shape
    if len(cam_ids) != n:
        raise RuntimeError("mtmc_mpn.build_graph: one camera id per node expected")
    if n > MAX_NODES:
        raise NotImplementedError(f"mtmc_mpn.build_graph: the Gram-matrix builder handles up to {MAX_NODES} nodes")
    dev = node_feats.device
    feats = node_feats if (node_feats.stride(1) == 1 and node_feats.stride(0) % 4 == 0) else node_feats.contiguous()
    in_list, in_off, out_list, out_off, block_off, n_cams = camera_tables(cam_ids)
    e = int(block_off[-1])
    lib = _lib.load()
    up = lambda a: torch.from_numpy(a).to(dev)
    t_in, t_inoff, t_out, t_outoff, t_blk = up(in_list), up(in_off), up(out_list), up(out_off), up(block_off)
    labels_dev = None
    if node_labels is not None:
        labels_dev = torch.as_tensor(np.asarray(node_labels), dtype=torch.int64).to(dev)
    x = torch.empty((n, f), dtype=torch.float32, device=dev)
    edge_pairs = torch.empty((e, 2), dtype=torch.int64, device=dev)
    edge_attr = torch.empty((e, 2), dtype=torch.float32, device=dev)
    edge_labels = torch.empty((e,), dtype=torch.float32, device=dev) if labels_dev is not None else None
    need = lib.mtmc_graph_workspace_bytes(n, f)
    if need == 0:
        raise RuntimeError("mtmc_mpn.build_graph: unsupported size")
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.mtmc_build_graph(
            feats.data_ptr(), feats.stride(0), n, f, 1 if l2norm else 0,
            t_in.data_ptr(), t_inoff.data_ptr(), t_out.data_ptr() if e else None, t_outoff.data_ptr(), t_blk.data_ptr(),
            n_cams, e, labels_dev.data_ptr() if labels_dev is not None else None,
            x.data_ptr(), edge_pairs.data_ptr() if e else None, edge_attr.data_ptr() if e else None,
            edge_labels.data_ptr() if edge_labels is not None else None,
            ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"mtmc_mpn.build_graph failed (code {rc})")
    return types.SimpleNamespace(x=x, edge_index=edge_pairs.t(), edge_attr=edge_attr, edge_labels=edge_labels,
                                 y=labels_dev)
