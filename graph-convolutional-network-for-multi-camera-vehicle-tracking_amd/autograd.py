"""Training path: `outputs, _ = model(data); loss.backward()` (reference train.py:356,424).

The autograd formula of the hot path lives with its op registration: `torch.ops.mtmc_mpn.mp_forward` records the tape
(training-mode HIP forward: Dropout masks from a counter-based generator, every round's buffers kept in a per-call
workspace) and its registered backward calls `torch.ops.mtmc_mpn.mp_backward`, which writes the gradients of all 34
parameters (and of x / edge_attr when asked) -- see torch_ops.py.  This module keeps the helpers older callers used.
"""
from __future__ import annotations

from . import torch_ops  # noqa: F401
from .engine import ordered_params as _ordered


def _ordered_params(engine):
    """Parameters in the order the flat list of `torch.ops.mtmc_mpn.mp_forward` expects them."""
    return _ordered(engine.module)


def forward_with_tape(engine, x, edge_index, edge_attr, training):
    mod = engine.module
    was = mod.training
    mod.train(bool(training))
    try:
        import types
        out, h = mod(types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr))
    finally:
        mod.train(was)
    return out["classified_edges"], h
