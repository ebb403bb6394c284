"""Seeded synthetic post-processing scenarios (classifier logits over a cross-camera graph).

Shared by the golden generator, the parity tests and bench.py, so that fixtures can hold recipes + hashes instead
of the arrays.  A scenario imitates what the MPN's last classifier emits on a tracklet graph: identities seen by a
random subset of the cameras, confident logits on same-identity pairs, a controllable rate of false positives /
false negatives per *directed* edge (so single-direction edges, over-full nodes and over-sized clusters all
occur), optionally quantised logits (probability ties, saturation at exactly 1.0)."""
from __future__ import annotations

import types

import torch

from mtmc_mpn import graphs


def scenario(n_ids: int = 30, n_cams: int = 4, seed: int = 0, p_seen: float = 0.8, fp_rate: float = 0.002,
             fn_rate: float = 0.05, sigma: float = 2.0, quant: float = 0.0, scale: float = 1.0,
             perm: bool = False, pair_fp: float = 0.0):
    g = torch.Generator().manual_seed(seed)
    seen = torch.rand(n_ids, n_cams, generator=g) < p_seen
    seen[torch.arange(n_ids), torch.randint(0, n_cams, (n_ids,), generator=g)] = True     # every identity somewhere
    cam, ident = [], []
    for c in range(n_cams):                       # nodes ordered camera by camera (libs/dataset.py:279-281)
        ids = torch.nonzero(seen[:, c]).flatten()
        cam.append(torch.full((ids.numel(),), c, dtype=torch.int64))
        ident.append(ids)
    cam, ident = torch.cat(cam), torch.cat(ident)
    edge_index = graphs.camera_edge_index(cam)
    if perm:
        edge_index = edge_index[:, torch.randperm(edge_index.shape[1], generator=g)]
    row, col = edge_index[0], edge_index[1]
    e = row.numel()
    same = ident[row] == ident[col]
    flip = torch.rand(e, generator=g) < torch.where(same, torch.tensor(fn_rate), torch.tensor(fp_rate))
    if pair_fp > 0:                               # false positives on BOTH directions of a pair (survive the cut)
        lo, hi = torch.minimum(row, col), torch.maximum(row, col)
        u = torch.rand(cam.numel(), cam.numel(), generator=g)
        flip = flip | (~same & (u[lo, hi] < pair_fp))
    positive = same ^ flip
    margin = torch.where(positive, torch.tensor(3.0), torch.tensor(-4.0)) + sigma * torch.randn(e, generator=g)
    margin = margin * scale
    if quant > 0:
        margin = torch.round(margin / quant) * quant
    base = torch.randn(e, generator=g)
    logits = torch.stack([base - 0.5 * margin, base + 0.5 * margin], dim=1).contiguous()
    return types.SimpleNamespace(edge_index=edge_index, logits=logits, n_nodes=int(cam.numel()), n_cams=n_cams,
                                 cam=cam, ident=ident)
