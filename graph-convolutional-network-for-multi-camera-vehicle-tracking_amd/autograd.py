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
        ctx.mark_non_differentiable(edge_index)
        # one output per classified step: autograd then hands each step's gradient over as it is (no zeros + slice-adds)
        return tuple(prep.logits[i] for i in range(prep.logits.shape[0])) + (prep.h,)

    @staticmethod
    def backward(ctx, *grads_out):
        engine, prep = ctx.engine, ctx.prep
        dev = prep.dev
        d_steps, d_h = grads_out[:-1], grads_out[-1]
        # all 34 parameter gradients carved from one buffer: one allocation, one memset in the library
        layers = list(engine.param_layers())
        sizes = []
        for _, lin, bn, _ in layers:
            sizes += [lin.weight.numel(), lin.bias.numel()]
            if bn is not None:
                sizes += [bn.weight.numel(), bn.bias.numel()]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 63) // 64 * 64                       # 256-byte aligned pieces
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grads_struct = _lib.Model()
        grad_tensors = []
        it = iter(offs)

        def piece(like):
            o = next(it)
            t = flat[o:o + like.numel()].view(like.shape)
            grad_tensors.append(t)
            return t.data_ptr()
        for (slot, idx), lin, bn, layer in layers:
            dst = getattr(grads_struct, slot) if idx is None else getattr(grads_struct, slot)[idx]
            dst.weight, dst.bias = piece(lin.weight), piece(lin.bias)
            if bn is not None:
                dst.gamma, dst.beta = piece(bn.weight), piece(bn.bias)
            dst.in_dim, dst.out_dim = layer.in_dim, layer.out_dim
        dx = torch.empty((prep.n, engine.spec.enc_node[0].in_dim), device=dev) if ctx.need_x else None
        dattr = torch.empty((prep.e, engine.spec.enc_edge[0].in_dim), device=dev) if ctx.need_attr else None
        keep = [g.contiguous().float() if g is not None else None for g in d_steps]
        steps = (C.c_void_p * max(len(keep), 1))(*[g.data_ptr() if g is not None and g.numel() else None for g in keep])
        dh = d_h.contiguous().float() if d_h is not None else None
        with torch.cuda.device(dev):
            prep.call.stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(engine.lib.mtmc_mpn_backward_steps(
                C.byref(prep.model), C.byref(prep.call), steps, dh.data_ptr() if dh is not None else None,
                C.byref(grads_struct), flat.data_ptr(), flat.numel() * 4,
                dx.data_ptr() if dx is not None else None, dattr.data_ptr() if dattr is not None else None))
        if engine.spec.num_enc_steps == 0:        # the update MLPs took no part: None, as autograd gives the reference
            dead = set()
            for (slot, _), lin, bn, _ in layers:
                if slot in ("upd_edge", "upd_node"):
                    dead |= {id(lin.weight), id(lin.bias)} | ({id(bn.weight), id(bn.bias)} if bn is not None else set())
            order = []
            for _, lin, bn, _ in layers:
                order += [lin.weight, lin.bias] + ([bn.weight, bn.bias] if bn is not None else [])
            grad_tensors = [None if id(p) in dead else g for p, g in zip(order, grad_tensors)]
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
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0    # follows torch.manual_seed
    out = _MpnFunction.apply(engine, bool(training), seed, x, edge_index, edge_attr, *_ordered_params(engine))
    return list(out[:-1]), out[-1]
