"""Training path: `outputs, _ = model(data); loss.backward()` (reference train.py:356,424).

`torch.autograd.Function` glue only: the forward is the training-mode HIP forward (Dropout masks from a
counter-based generator, every round's buffers kept in a per-call workspace = the tape), the backward is
`mtmc_mpn_backward`, which writes the gradients of all 34 parameters (and of x / edge_attr when asked).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _MpnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, training, seed, x, edge_index, edge_attr, *params):
        prep = engine.prepare(x.detach(), edge_index, edge_attr.detach(), tape=True, seed=seed)
        if not training:                       # eval-mode statistics/identity Dropout, but still differentiable
            prep.model.dropout_enc = prep.model.dropout_upd_edge = prep.model.dropout_upd_node = 0.0
        with torch.cuda.device(prep.dev):
            _lib.check(engine.lib.mtmc_mpn_forward(C.byref(prep.model), C.byref(prep.call)))
        ctx.engine, ctx.prep = engine, prep
        ctx.need_x, ctx.need_attr = x.requires_grad, edge_attr.requires_grad
        ctx.param_shapes = [p.shape for p in params]
        ctx.mark_non_differentiable(edge_index)
        return prep.logits, prep.h

    @staticmethod
    def backward(ctx, d_logits, d_h):
        engine, prep = ctx.engine, ctx.prep
        dev = prep.dev
        grads_struct = _lib.Model()
        grad_tensors = []
        for (slot, idx), lin, bn, layer in engine.param_layers():
            dst = getattr(grads_struct, slot) if idx is None else getattr(grads_struct, slot)[idx]
            gw, gb = torch.empty_like(lin.weight), torch.empty_like(lin.bias)
            dst.weight, dst.bias = gw.data_ptr(), gb.data_ptr()
            grad_tensors += [gw, gb]
            if bn is not None:
                gg, gt = torch.empty_like(bn.weight), torch.empty_like(bn.bias)
                dst.gamma, dst.beta = gg.data_ptr(), gt.data_ptr()
                grad_tensors += [gg, gt]
            dst.in_dim, dst.out_dim = layer.in_dim, layer.out_dim
        dx = torch.empty((prep.n, engine.spec.enc_node[0].in_dim), device=dev) if ctx.need_x else None
        dattr = torch.empty((prep.e, engine.spec.enc_edge[0].in_dim), device=dev) if ctx.need_attr else None
        dl = d_logits.contiguous().float() if d_logits is not None else None
        dh = d_h.contiguous().float() if d_h is not None else None
        with torch.cuda.device(dev):
            prep.call.stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(engine.lib.mtmc_mpn_backward(
                C.byref(prep.model), C.byref(prep.call), dl.data_ptr() if dl is not None and dl.numel() else None,
                dh.data_ptr() if dh is not None else None, C.byref(grads_struct),
                dx.data_ptr() if dx is not None else None, dattr.data_ptr() if dattr is not None else None))
        return (None, None, None, dx, None, dattr) + tuple(grad_tensors)


def _ordered_params(engine):
    """Parameters in the order `_MpnFunction.backward` returns their gradients."""
    out = []
    for _, lin, bn, _ in engine.param_layers():
        out += [lin.weight, lin.bias]
        if bn is not None:
            out += [bn.weight, bn.bias]
    return out


def forward_with_tape(engine, x, edge_index, edge_attr, training):
    if engine.spec.num_enc_steps < 1:
        raise NotImplementedError("mtmc_mpn: backward with num_enc_steps == 0 is not implemented")
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0    # follows torch.manual_seed
    logits, h = _MpnFunction.apply(engine, bool(training), seed, x, edge_index, edge_attr, *_ordered_params(engine))
    n_out = logits.shape[0]
    return [logits[i] for i in range(n_out)], h
