"""`postprocess`: the callers' post-processing of the MPN logits on the GPU.

Replaces what reference inference.py:475-489 and `post_processing` (inference.py:70-169, utils.py:30-339) do after the
MPN call -- softmax / argmax, the D2H copy of the edge list, the Python list scans, networkx SCC per iteration -- by one
library call (`mtmc_postprocess`).  Same `predictions`, same cluster numbering `ID_pred`.
"""
from __future__ import annotations

import types

import torch

from . import _lib

CUTTING, PRUNING, SPLITTING = 1, 2, 4
_STATUS = {1: "more active edges than max_active: predictions left at argmax",
           2: "over-sized cluster without an active edge",
           3: "splitting did not finish within 262144 iterations (predictions hold the state reached)",
           4: "edge_index holds node ids outside [0, num_nodes) on an active edge (clamped on the device)"}


def _flag(v) -> bool:
    """CONFIG['CUTTING'] etc. arrive as bool or as the strings 'True' / 'False' (inference.py:75-91)."""
    return v == "True" if isinstance(v, str) else bool(v)


def postprocess(logits, edge_index, num_nodes: int, num_cameras: int, cutting=True, pruning=True, splitting=True,
                preds_prob=None, predictions=None, max_active: int = 0, check: bool = True):
    """logits: [E, 2] float32 on a ROCm GPU (`outputs['classified_edges'][-1]`); edge_index: [2, E] int64 (any
    strides).  Returns a namespace with ID_pred [N] int64, predictions [E] int64, preds_prob1 [E] float32 (all on
    the device) and `info` (dict of counters; read with one small D2H copy when `check`).
    Pass `preds_prob` ([E,2] or [E]) and `predictions` instead of logits (`logits=None`) to post-process given
    probabilities exactly."""
    ref = logits if logits is not None else predictions
    if not (isinstance(ref, torch.Tensor) and ref.is_cuda):
        raise RuntimeError("mtmc_mpn.postprocess: tensors must be on a ROCm GPU (no CPU path)")
    dev = ref.device
    e = int(edge_index.shape[1])
    if edge_index.dtype != torch.int64 or edge_index.shape[0] != 2 or edge_index.device != dev:
        raise RuntimeError("mtmc_mpn.postprocess: edge_index must be int64 [2, E] on the same device")
    row, col = edge_index[0], edge_index[1]
    if row.stride(0) != col.stride(0) or row.stride(0) < 1:
        edge_index = edge_index.contiguous()
        row, col = edge_index[0], edge_index[1]
    if logits is not None:
        if logits.dtype != torch.float32 or tuple(logits.shape) != (e, 2):
            raise RuntimeError("mtmc_mpn.postprocess: logits must be float32 [E, 2]")
        logits = logits.contiguous()
        prob1 = torch.empty(e, dtype=torch.float32, device=dev)
        pred = torch.empty(e, dtype=torch.int64, device=dev)
    else:
        if preds_prob is None or predictions is None:
            raise RuntimeError("mtmc_mpn.postprocess: give logits, or preds_prob and predictions")
        prob1 = (preds_prob[:, 1] if preds_prob.dim() == 2 else preds_prob).to(torch.float32).contiguous().clone()
        pred = predictions.to(torch.int64).contiguous().clone()
    ids = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    info = torch.zeros(8, dtype=torch.int32, device=dev)
    lib = _lib.load()
    need = lib.mtmc_postprocess_workspace_bytes(num_nodes, e, max_active)
    if need == 0:
        raise RuntimeError("mtmc_mpn.postprocess: unsupported size")
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    flags = (CUTTING if _flag(cutting) else 0) | (PRUNING if _flag(pruning) else 0) | (SPLITTING if _flag(splitting) else 0)
    with torch.cuda.device(dev):
        rc = lib.mtmc_postprocess(logits.data_ptr() if logits is not None else None, row.data_ptr() if e else None,
                                  col.data_ptr() if e else None, row.stride(0) if e else 1, num_nodes, e, num_cameras,
                                  flags, max_active, prob1.data_ptr(), pred.data_ptr(), ids.data_ptr(), info.data_ptr(),
                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"mtmc_mpn.postprocess failed (code {rc})")
    out = types.SimpleNamespace(ID_pred=ids, predictions=pred, preds_prob1=prob1, info_dev=info, info=None)
    if check:
        v = info.cpu().tolist()
        out.info = dict(active_in=v[0], active_out=v[1], clusters=v[2], status=v[3], split_iterations=v[4],
                        component_walks=v[5], pruning_rounds=v[6], walk_us=v[7] / 100.0)
        if v[3]:
            raise RuntimeError("mtmc_mpn.postprocess: " + _STATUS.get(v[3], f"status {v[3]}"))
    return out
