"""Stand-alone ops of the hot path, for callers that use the pieces on their own:

  scatter_add / scatter_mean / scatter_max   the call forms of the third-party `torch_scatter` op the
      reference aggregates with (`scatter_*(src, index, dim=0, dim_size=N)`, reference models/mpn.py:196-202)
  mlp_forward                                `MLP.forward` (reference models/mlp.py:32-33), eval mode
  cross_entropy                              `F.cross_entropy(input, target, weight, reduction=...)` on the [E, C<=4]
      edge logits, forward and backward fused (the training callers' loss, reference train.py:88-93, :109-142)
  edge_confusion                             TP / FP / TN / FN of argmax(logits) against the edge labels in one pass,
      without the device synchronisations of boolean-mask indexing (reference train.py:98-107, inference.py:20-67)

All of them run HIP kernels through the C ABI; CPU tensors are refused.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.mtmc_mpn.*)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _check_scatter(src, index, out, name):
    if out is not None:
        raise NotImplementedError(f"mtmc_mpn.{name}: `out=` is not supported")
    if not (src.is_cuda and index.is_cuda):
        raise RuntimeError("mtmc_mpn.scatter_*: tensors must be on a ROCm GPU (no CPU path)")


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    """`torch_scatter.scatter_add(src, index, dim=0, dim_size=N)`; integer `src` (the 1-D int64 form of reference
    utils.py:173-174) is summed exactly and keeps its dtype."""
    _check_scatter(src, index, out, "scatter_add")
    return torch.ops.mtmc_mpn.scatter_add(src, index, dim, dim_size)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    _check_scatter(src, index, out, "scatter_mean")
    return torch.ops.mtmc_mpn.scatter_mean(src, index, dim, dim_size)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    """Returns (values, argmax) like torch_scatter; rows nobody writes hold 0 and argmax = src.size(0)."""
    _check_scatter(src, index, out, "scatter_max")
    return torch.ops.mtmc_mpn.scatter_max(src, index, dim, dim_size)


def layer_forward(a: torch.Tensor, weight, bias, gamma=None, beta=None) -> torch.Tensor:
    """One Linear (+ BatchNorm1d with batch statistics + ReLU when gamma/beta are given) on the GPU."""
    if gamma is not None and a.shape[0] < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size {}".format(list(a.shape)))
    a = a.float()
    if a.stride(-1) != 1:
        a = a.contiguous()
    out_dim, in_dim = weight.shape
    lay = _lib.Layer()
    lay.weight, lay.bias = weight.data_ptr(), bias.data_ptr()
    lay.gamma = gamma.data_ptr() if gamma is not None else None
    lay.beta = beta.data_ptr() if beta is not None else None
    lay.in_dim, lay.out_dim = in_dim, out_dim
    y = torch.empty((a.shape[0], out_dim), dtype=torch.float32, device=a.device)
    stats = torch.empty((2 * out_dim,), dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().mtmc_mlp_layer_forward(C.byref(lay), a.data_ptr(), a.stride(0), a.shape[0], y.data_ptr(),
                                                      stats.data_ptr(), _stream(a.device)))
    return y


def mlp_forward(mlp, inp: torch.Tensor) -> torch.Tensor:
    """Eval-mode `MLP.forward`: per hidden layer Linear -> BatchNorm1d (batch statistics) -> ReLU."""
    if not inp.is_cuda:
        raise RuntimeError("mtmc_mpn.MLP: input must be on a ROCm GPU (no CPU path)")
    if mlp.training and any(l.dropout_p for l in mlp.layers):
        raise NotImplementedError("mtmc_mpn.MLP: training-mode Dropout is not built yet; use .eval()")
    if torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in mlp.parameters())):
        raise NotImplementedError("mtmc_mpn.MLP: backward is not built yet; call under torch.no_grad()")
    a = inp
    for spec in mlp.layers:
        lin = mlp.fc_layers[spec.lin_slot]
        bn = mlp.fc_layers[spec.bn_slot] if spec.bn_slot is not None else None
        a = layer_forward(a, lin.weight, lin.bias, bn.weight if bn is not None else None,
                          bn.bias if bn is not None else None)
        if spec.relu and bn is None:
            a = torch.relu_(a)
    return a


_REDUCTIONS = {"mean": 0, "sum": 1, "none": 2}


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, mode, ignore_index):
        n, c = logits.shape
        dev = logits.device
        x = logits.contiguous()
        sums = torch.empty(2 * _lib.STAT_REPLICAS, dtype=torch.float64, device=dev)
        per = torch.empty(n, dtype=torch.float32, device=dev) if mode == 2 else None
        loss = torch.empty((), dtype=torch.float32, device=dev) if mode != 2 else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtmc_cross_entropy_forward(
                x.data_ptr(), target.data_ptr(), weight.data_ptr() if weight is not None else None, n, c, ignore_index,
                mode, per.data_ptr() if per is not None else None, sums.data_ptr(),
                loss.data_ptr() if loss is not None else None, _stream(dev)))
        ctx.save_for_backward(x, target, weight, sums)
        ctx.mode, ctx.ignore_index = mode, ignore_index
        return per if mode == 2 else loss

    @staticmethod
    def backward(ctx, grad):
        x, target, weight, sums = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        d = torch.empty_like(x)
        g = grad.contiguous().float()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtmc_cross_entropy_backward(
                x.data_ptr(), target.data_ptr(), weight.data_ptr() if weight is not None else None, n, c,
                ctx.ignore_index, ctx.mode, g.data_ptr(), sums.data_ptr(), d.data_ptr(), _stream(dev)))
        return d, None, None, None, None


def cross_entropy(input, target, weight=None, reduction: str = "mean", ignore_index: int = -100):
    """Drop-in for `torch.nn.functional.cross_entropy(input, target, weight=weight, reduction=reduction)` with class
    indices as targets, for the [E, C<=4] float32 edge logits of the MPN."""
    if not (input.is_cuda and target.is_cuda):
        raise RuntimeError("mtmc_mpn.cross_entropy: tensors must be on a ROCm GPU (no CPU path)")
    if input.dim() != 2 or input.dtype != torch.float32 or input.shape[1] > 4 or target.shape != input.shape[:1]:
        raise NotImplementedError("mtmc_mpn.cross_entropy: input must be float32 [E, C<=4], target int64 [E]")
    if reduction not in _REDUCTIONS:
        raise ValueError(f"{reduction} is not a valid value for reduction")
    if weight is not None:
        weight = weight.to(device=input.device, dtype=torch.float32).contiguous()
        if weight.numel() != input.shape[1]:
            raise RuntimeError("mtmc_mpn.cross_entropy: weight must have one entry per class")
    return _CrossEntropy.apply(input, target.contiguous().long(), weight, _REDUCTIONS[reduction], int(ignore_index))


class _CrossEntropySteps(torch.autograd.Function):
    """The classified steps of one forward are consecutive [E, C] slices of one block: one pass over all of them."""

    @staticmethod
    def forward(ctx, target, weight, mode, ignore_index, *steps):
        n, c = steps[0].shape
        dev = steps[0].device
        sums = torch.empty(2 * _lib.STAT_REPLICAS, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtmc_cross_entropy_steps_forward(
                steps[0].data_ptr(), target.data_ptr(), weight.data_ptr() if weight is not None else None, n, c,
                len(steps), ignore_index, mode, sums.data_ptr(), loss.data_ptr(), _stream(dev)))
        ctx.save_for_backward(target, weight, sums, *steps)
        ctx.mode, ctx.ignore_index = mode, ignore_index
        return loss

    @staticmethod
    def backward(ctx, grad):
        target, weight, sums, *steps = ctx.saved_tensors
        n, c = steps[0].shape
        dev = steps[0].device
        d = torch.empty((len(steps), n, c), dtype=torch.float32, device=dev)
        g = grad.contiguous().float()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtmc_cross_entropy_steps_backward(
                steps[0].data_ptr(), target.data_ptr(), weight.data_ptr() if weight is not None else None, n, c,
                len(steps), ctx.ignore_index, ctx.mode, g.data_ptr(), sums.data_ptr(), d.data_ptr(), _stream(dev)))
        return (None, None, None, None) + tuple(d[i] for i in range(len(steps)))


def cross_entropy_steps(steps, target, weight=None, reduction: str = "mean", ignore_index: int = -100):
    """`sum(cross_entropy(step, target, weight, reduction) for step in steps)` -- the training loop's loss over
    `outputs['classified_edges']` (reference train.py:118-138) -- in ONE pass when the steps are the consecutive slices
    of the logits block a forward of this package returns (4 launches instead of 4 per step); any other list of
    tensors is summed step by step."""
    steps = list(steps)
    if not steps:
        raise ValueError("cross_entropy_steps: no classified steps")
    if reduction not in ("mean", "sum"):
        raise ValueError("cross_entropy_steps: reduction must be 'mean' or 'sum'")
    s0 = steps[0]
    block = (s0.dim() == 2 and s0.dtype == torch.float32 and s0.is_cuda and target.is_cuda and s0.shape[1] <= 4
             and target.shape == s0.shape[:1]
             and all(t.shape == s0.shape and t.dtype == s0.dtype and t.device == s0.device and t.is_contiguous()
                     and t.data_ptr() == s0.data_ptr() + i * s0.numel() * 4 for i, t in enumerate(steps)))
    if not block or len(steps) == 1:
        return sum(cross_entropy(t, target, weight=weight, reduction=reduction, ignore_index=ignore_index) for t in steps)
    if weight is not None:
        weight = weight.to(device=s0.device, dtype=torch.float32).contiguous()
        if weight.numel() != s0.shape[1]:
            raise RuntimeError("mtmc_mpn.cross_entropy_steps: weight must have one entry per class")
    return _CrossEntropySteps.apply(target.contiguous().long(), weight, _REDUCTIONS[reduction], int(ignore_index), *steps)


def edge_confusion(logits, labels):
    """int64 tensor [TP, FP, TN, FN] (on the device) of `argmax(logits, 1) == 1` against 0/1 `labels`; e.g.
    FPR = FP / (FP + TN) as in train.py:100-102.  No host synchronisation."""
    if not (logits.is_cuda and labels.is_cuda):
        raise RuntimeError("mtmc_mpn.edge_confusion: tensors must be on a ROCm GPU (no CPU path)")
    if logits.dim() != 2 or logits.dtype != torch.float32 or not 2 <= logits.shape[1] <= 4 or labels.shape != logits.shape[:1]:
        raise NotImplementedError("mtmc_mpn.edge_confusion: logits must be float32 [E, 2..4], labels [E]")
    x, y = logits.contiguous(), labels.contiguous().long()
    out = torch.empty(4, dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mtmc_edge_confusion(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(),
                                                   _stream(x.device)))
    return out
