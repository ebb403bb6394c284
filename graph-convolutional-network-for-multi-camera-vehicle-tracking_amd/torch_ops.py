"""PyTorch-ROCm custom-op registration of the hot path: `torch.ops.mtmc_mpn.*`.

BASELINE.json's north_star asks for the MPN forward "registered as a PyTorch-ROCm custom op so main.py / train.py /
inference.py call an unchanged MPN nn.Module signature"; SURVEY.md 8(b) names the namespace.  The ops below are thin
schemas over the C ABI of include/mtmc_mpn.h (one library call each, on the current HIP stream) -- there is no second
implementation behind them: `MOTMPNet.forward` itself dispatches through `torch.ops.mtmc_mpn.mp_forward`, autograd
through `torch.ops.mtmc_mpn.mp_backward`.  CUDA (= ROCm) dispatch key only: CPU tensors have no kernel and raise.

  mp_forward(x, edge_index, edge_attr, params, config, training, seed, flags, tape) -> (logits[S,E,C], h[N,32], tape)
      the whole `MOTMPNet.forward` (reference models/mpn.py:250-299).  `params`: the 34 tensors in struct order
      (`engine.layer_slots`: node encoder, edge encoder, edge update, node update, classifier; weight, bias, gamma, beta),
      `config`: JSON of {"params": GRAPH_NET_PARAMS, "arch": ...}.  `tape=True` keeps every round's buffers in the third
      output for `mp_backward` (training / grad mode); otherwise it is empty.
  mp_backward(tape, x, edge_index, edge_attr, params, config, training, seed, flags, d_logits, d_h, need_x, need_attr)
      -> (flat parameter gradients, d_x, d_edge_attr)         (reference train.py:424, loss.backward())
  encode_nodes(x, params, config) -> h0[N,32]                 the node encoder alone (models/mpn.py:131, eval mode)
  scatter_add / scatter_mean / scatter_max(src, index, dim, dim_size)   the torch_scatter call forms
      (models/mpn.py:196-202; the 1-D int64 form of utils.py:173-174 returns int64)
"""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from . import config as _config

CHECK_INDICES = 1 << 16          # host-side flag bit of `flags`: synchronise and raise IndexError on bad edge_index
NO_WEIGHT_CACHE = 1 << 17        # host-side: no weight-plane cache for this call (engine.weight_plane_cache)
_ENGINES = {}


def config_key(model_params, arch) -> str:
    return json.dumps({"params": model_params, "arch": arch}, sort_keys=True, default=str)


def engine_for(config: str):
    eng = _ENGINES.get(config)
    if eng is None:
        cfg = json.loads(config)
        spec = _config.resolve(cfg["params"], cfg["arch"])
        ok, why = _config.check_supported(spec)
        if not ok:
            raise NotImplementedError("mtmc_mpn HIP path does not cover this GRAPH_NET_PARAMS: " + why)
        from .engine import ForwardEngine
        eng = _ENGINES[config] = ForwardEngine(spec)
    return eng


_GRAD_LAYOUTS = {}


def grad_layout(spec):
    """(offset, numel, shape) of every parameter gradient inside the flat buffer mp_backward returns (256-byte pieces;
    the rule of mtmc_mpn_grad_layout in csrc/api_train.hip)."""
    key = tuple((slot, idx, layer.in_dim, layer.out_dim, layer.bn_slot is not None)
                for slot, idx, layer in _layer_slots(spec))      # by content: id(spec) can be reused after a collection
    hit = _GRAD_LAYOUTS.get(key)
    if hit is not None:
        return hit
    res = _grad_layout(spec)
    _GRAD_LAYOUTS[key] = res
    return res


def _grad_views(flat, layout):
    """The 34 gradient tensors as views of the flat buffer: one as_strided each (a slice + a view per tensor was 68 dispatcher
    calls per backward, a tenth of the launch-by-launch step's host time on a slow host)."""
    base = flat.storage_offset()
    return [flat.as_strided(shp, (shp[1], 1) if len(shp) == 2 else (1,), base + o) for o, _, shp in layout]


def _layer_slots(spec):
    from .engine import layer_slots
    return layer_slots(spec)


def _grad_layout(spec):
    from .engine import layer_slots
    out, total = [], 0
    for _, _, layer in layer_slots(spec):
        shapes = [(layer.out_dim, layer.in_dim), (layer.out_dim,)]
        if layer.bn_slot is not None:
            shapes += [(layer.out_dim,), (layer.out_dim,)]
        for shp in shapes:
            n = 1
            for d in shp:
                n *= d
            out.append((total, n, shp))
            total += (n + 63) // 64 * 64
    return out, total


# ---------------------------------------------------------------------------------------------------------------
# schemas
# ---------------------------------------------------------------------------------------------------------------
torch.library.define(
    "mtmc_mpn::mp_forward",
    "(Tensor x, Tensor edge_index, Tensor edge_attr, Tensor[] params, str config, bool training, int seed, int flags, "
    "bool tape) -> (Tensor, Tensor, Tensor)")
torch.library.define(
    "mtmc_mpn::mp_backward",
    "(Tensor(a!) tape, Tensor x, Tensor edge_index, Tensor edge_attr, Tensor[] params, str config, bool training, int seed, "
    "int flags, Tensor? d_logits, Tensor? d_h, bool need_x, bool need_attr) -> (Tensor, Tensor, Tensor)")
torch.library.define("mtmc_mpn::encode_nodes", "(Tensor x, Tensor[] params, str config) -> Tensor")
torch.library.define("mtmc_mpn::scatter_add", "(Tensor src, Tensor index, int dim, int? dim_size) -> Tensor")
torch.library.define("mtmc_mpn::scatter_mean", "(Tensor src, Tensor index, int dim, int? dim_size) -> Tensor")
torch.library.define("mtmc_mpn::scatter_max", "(Tensor src, Tensor index, int dim, int? dim_size) -> (Tensor, Tensor)")


def _n_out(spec):
    return min(spec.num_class_steps, spec.num_enc_steps) if spec.num_enc_steps > 0 else 1


# ---------------------------------------------------------------------------------------------------------------
# the forward
# ---------------------------------------------------------------------------------------------------------------
def _forward_prep(x, edge_index, edge_attr, params, config, training, seed, flags, tape):
    """One forward: returns the prepared call (structs, outputs, workspace = tape)."""
    eng = engine_for(config)
    eng.flags = int(flags) & 0xFFFF
    eng.weight_cache = not (int(flags) & NO_WEIGHT_CACHE)
    prep = eng.prepare(x, edge_index, edge_attr, tape=bool(tape), seed=seed, params=list(params))
    if tape and not training:                  # eval-mode statistics / identity Dropout, but still differentiable
        prep.model.dropout_enc = prep.model.dropout_upd_edge = prep.model.dropout_upd_node = 0.0
    with torch.cuda.device(prep.dev):
        _lib.check(eng.lib.mtmc_mpn_forward(C.byref(prep.model), C.byref(prep.call)))
        if flags & CHECK_INDICES:
            lay = eng.layout(prep)
            bits = prep.ws[lay.flags_off:lay.flags_off + 32].view(torch.int32).cpu()
            if int(bits[1]) != 0:
                raise IndexError("mtmc_mpn: edge_index holds node ids outside [0, N)")
    return prep


def _mp_forward(x, edge_index, edge_attr, params, config, training, seed, flags, tape):
    prep = _forward_prep(x, edge_index, edge_attr, params, config, training, seed, flags, tape)
    return prep.logits, prep.h, (prep.ws if tape else prep.ws.new_empty(0))


torch.library.impl("mtmc_mpn::mp_forward", "CUDA")(_mp_forward)   # (call form: the decorator does not return the function)


@torch.library.register_fake("mtmc_mpn::mp_forward")
def _mp_forward_fake(x, edge_index, edge_attr, params, config, training, seed, flags, tape):
    eng = engine_for(config)
    spec = eng.spec
    e = edge_index.shape[1]
    tape_bytes = 0
    if tape:                                   # the real op returns the training workspace: size it from the dimensions
        model = eng.shape_model()
        tape_bytes = int(eng.lib.mtmc_mpn_train_workspace_bytes(C.byref(model), int(x.shape[0]), int(e))) + 256
    return (x.new_empty((_n_out(spec), e, spec.cls_edge[0].out_dim)), x.new_empty((x.shape[0], spec.node_dim)),
            x.new_empty((tape_bytes,), dtype=torch.uint8))


def _mp_backward(tape, x, edge_index, edge_attr, params, config, training, seed, flags, d_logits, d_h, need_x, need_attr,
                 prep=None, d_steps=None):
    """prep: the forward's prepared call (the lean autograd path keeps it: same tensors, same structs); None: rebuilt.
    d_steps: the gradients of the classified steps as separate [E, C] tensors (None entries: no gradient), instead of d_logits."""
    eng = engine_for(config)
    eng.flags = int(flags) & 0xFFFF
    spec = eng.spec
    if prep is None:
        prep = eng.prepare(x, edge_index, edge_attr, tape=True, seed=seed, params=list(params), tape_ws=tape)
        if not training:
            prep.model.dropout_enc = prep.model.dropout_upd_edge = prep.model.dropout_upd_node = 0.0
    dev = prep.dev
    _, total = grad_layout(spec)                          # == mtmc_mpn_grad_layout (tests/test_torch_ops_registration.py)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    dx = torch.empty((prep.n, spec.enc_node[0].in_dim), device=dev) if need_x else flat.new_empty(0)
    dattr = torch.empty((prep.e, spec.enc_edge[0].in_dim), device=dev) if need_attr else flat.new_empty(0)
    dl = d_logits.contiguous().float() if d_logits is not None else None
    dh = d_h.contiguous().float() if d_h is not None else None
    n_steps = _n_out(spec)
    step_bytes = 4 * prep.e * spec.cls_edge[0].out_dim
    if d_steps is not None:
        d_steps = [t.contiguous().float() if t is not None and t.numel() else None for t in d_steps]   # (kept alive to the call)
        steps = (C.c_void_p * max(n_steps, 1))(*[t.data_ptr() if t is not None else None for t in d_steps])
    else:
        steps = (C.c_void_p * max(n_steps, 1))(*[(dl.data_ptr() + i * step_bytes) if dl is not None and dl.numel() else None
                                                 for i in range(n_steps)])
    with torch.cuda.device(dev):
        prep.call.stream = _stream(dev)
        _lib.check(eng.lib.mtmc_mpn_backward_flat(
            C.byref(prep.model), C.byref(prep.call), steps, dh.data_ptr() if dh is not None else None,
            flat.data_ptr(), flat.numel(), dx.data_ptr() if need_x else None, dattr.data_ptr() if need_attr else None))
    return flat, dx, dattr


torch.library.impl("mtmc_mpn::mp_backward", "CUDA")(_mp_backward)


@torch.library.register_fake("mtmc_mpn::mp_backward")
def _mp_backward_fake(tape, x, edge_index, edge_attr, params, config, training, seed, flags, d_logits, d_h, need_x, need_attr):
    _, total = grad_layout(engine_for(config).spec)
    return (x.new_empty((total,)), x.new_empty(x.shape if need_x else (0,)),
            edge_attr.new_empty(edge_attr.shape if need_attr else (0,)))


def _setup_context(ctx, inputs, output):
    x, edge_index, edge_attr, params, config, training, seed, flags, tape = inputs
    ctx.config, ctx.training, ctx.seed, ctx.flags, ctx.n_params = config, training, seed, flags, len(params)
    ctx.need_x, ctx.need_attr = x.requires_grad, edge_attr.requires_grad
    ctx.has_tape = bool(tape)
    ctx.save_for_backward(output[2], x, edge_index, edge_attr, *params)


def _autograd_backward(ctx, d_logits, d_h, _d_tape):
    if not ctx.has_tape:
        raise RuntimeError("mtmc_mpn: mp_forward was called with tape=False; nothing to differentiate through")
    tape, x, edge_index, edge_attr, *params = ctx.saved_tensors
    # mp_backward declares the tape mutable (it clears and fills its backward scratch regions; the forward's saved
    # activations are only read).  Autograd checks a saved tensor's version on every unpack, so a second backward over the
    # same graph (retain_graph=True) would refuse the tape after the first one bumped it: hand the op an alias with its own
    # version counter.  A repeated backward recomputes the same gradients (tests/test_gpu_training.py).
    flat, dx, dattr = torch.ops.mtmc_mpn.mp_backward(tape.data, x, edge_index, edge_attr, params, ctx.config, ctx.training,
                                                      ctx.seed, ctx.flags, d_logits, d_h, ctx.need_x, ctx.need_attr)
    spec = engine_for(ctx.config).spec
    layout, _ = grad_layout(spec)
    grads = _grad_views(flat, layout)
    if spec.num_enc_steps == 0:               # the update MLPs took no part: None, as autograd gives the reference
        from .engine import layer_slots
        i = 0
        for slot, _, layer in layer_slots(spec):
            k = 4 if layer.bn_slot is not None else 2
            if slot in ("upd_edge", "upd_node"):
                grads[i:i + k] = [None] * k
            i += k
    return (dx if ctx.need_x else None, None, dattr if ctx.need_attr else None, grads, None, None, None, None, None)


torch.library.register_autograd("mtmc_mpn::mp_forward", _autograd_backward, setup_context=_setup_context)


class _MpForwardLean(torch.autograd.Function):
    """The same forward / backward pair as the registered op's autograd formula -- `_mp_forward` and `_mp_backward` are the
    functions registered as its kernels -- called without the dispatcher around them.  With 34 parameter tensors in a
    `Tensor[]` argument torch.library's autograd wrapper (pytree flattening, a second Function.apply, redispatch) costs the
    host about 0.2 ms per forward and as much again per backward (tools/train_cpu_profile.py), on a training step whose
    kernels take 0.7 ms: `MOTMPNet.forward` uses this path under grad mode (reference train.py:356,424); eval-mode calls and
    every functional user go through `torch.ops.mtmc_mpn.mp_forward`."""

    @staticmethod
    def forward(ctx, x, edge_index, edge_attr, config, training, seed, flags, *params):
        prep = _forward_prep(x, edge_index, edge_attr, params, config, training, seed, flags, True)
        ctx.config, ctx.training, ctx.seed, ctx.flags = config, training, seed, flags
        ctx.need_x, ctx.need_attr = x.requires_grad, edge_attr.requires_grad
        # What the backward needs and NOTHING that refers to the outputs: round 4 kept `prep` itself here, whose logits / h are
        # this node's outputs -- a reference cycle through C++ (outputs -> grad_fn -> ctx -> prep -> outputs) that only Python's
        # generational GC could break, so several steps' tapes stayed resident.  The call structs hold raw
        # pointers only; the tape and the tensors those pointers name (prepare() may have made contiguous copies) are SAVED
        # tensors: released by autograd when the backward has run (kept under retain_graph), like any other saved activation.
        ctx.structs = (prep.model, prep.call, prep.n, prep.e, prep.dev)
        ctx.n_keep = len(prep.keep)
        ctx.save_for_backward(prep.ws, *prep.keep, *params)
        # One output per classified step (consecutive [E, C] slices of one block, which is what ops.cross_entropy_steps looks
        # for): the loss hands back one gradient per step and the library takes them as they are -- `logits.unbind(0)` outside
        # would make autograd stack them again (a 4 MB copy kernel per step at config 3).  An output nobody used (h, in every
        # training loop of the reference) gets None, not a zero-filled tensor.
        ctx.set_materialize_grads(False)
        return (*prep.logits.unbind(0), prep.h)

    @staticmethod
    def backward(ctx, *d_out):
        import types
        d_steps, d_h = d_out[:-1], d_out[-1]
        ws, *rest = ctx.saved_tensors
        keep, params = rest[:ctx.n_keep], rest[ctx.n_keep:]
        model, call, n, e, dev = ctx.structs
        prep = types.SimpleNamespace(model=model, call=call, ws=ws, n=n, e=e, dev=dev, keep=tuple(keep))
        x, edge_index, edge_attr = keep[0], keep[1], keep[2]
        flat, dx, dattr = _mp_backward(ws, x, edge_index, edge_attr, params, ctx.config, ctx.training, ctx.seed, ctx.flags,
                                       None, d_h, ctx.need_x, ctx.need_attr, prep=prep, d_steps=d_steps)
        spec = engine_for(ctx.config).spec
        layout, _ = grad_layout(spec)
        grads = _grad_views(flat, layout)
        if spec.num_enc_steps == 0:           # the update MLPs took no part: None, as autograd gives the reference
            i = 0
            for slot, _, layer in _layer_slots(spec):
                k = 4 if layer.bn_slot is not None else 2
                if slot in ("upd_edge", "upd_node"):
                    grads[i:i + k] = [None] * k
                i += k
        return (dx if ctx.need_x else None, None, dattr if ctx.need_attr else None, None, None, None, None, *grads)


def mp_forward_lean(x, edge_index, edge_attr, params, config, training, seed, flags):
    """([logits_step [E,C], ...], h [N,32]) with autograd, for CUDA tensors: see _MpForwardLean."""
    *steps, h = _MpForwardLean.apply(x, edge_index, edge_attr, config, training, seed, flags, *params)
    return steps, h


# ---------------------------------------------------------------------------------------------------------------
# the node encoder alone, and the torch_scatter call forms
# ---------------------------------------------------------------------------------------------------------------
@torch.library.impl("mtmc_mpn::encode_nodes", "CUDA")
def _encode_nodes(x, params, config):
    from . import ops
    from .engine import _check_param
    eng = engine_for(config)
    layers = eng.spec.enc_node
    if x.dim() != 2 or x.shape[1] != layers[0].in_dim or len(params) != 4 * len(layers):
        raise RuntimeError("mtmc_mpn.encode_nodes: x must be [N, in_dim] and params the 4 tensors of every encoder layer")
    if x.dtype != torch.float32:
        raise RuntimeError("mtmc_mpn.encode_nodes: x must be float32")
    # every pointer below goes to the kernels as is: device, dtype, contiguity and shape are checked HERE (a CPU, fp16,
    # strided or short tensor would be a wild read on the GPU)
    for i, layer in enumerate(layers):
        w, b, g, beta = params[4 * i:4 * i + 4]
        for t in (w, b, g, beta):
            _check_param(t, x.device)
        if tuple(w.shape) != (layer.out_dim, layer.in_dim) or any(tuple(t.shape) != (layer.out_dim,) for t in (b, g, beta)):
            raise RuntimeError(f"mtmc_mpn.encode_nodes: parameter shapes of encoder layer {i} do not match the configuration "
                               f"([{layer.out_dim}, {layer.in_dim}] weight, [{layer.out_dim}] bias / gamma / beta)")
    a = x
    for i, layer in enumerate(layers):
        w, b, g, beta = params[4 * i:4 * i + 4]
        a = ops.layer_forward(a, w, b, g, beta)
    return a


@torch.library.register_fake("mtmc_mpn::encode_nodes")
def _encode_nodes_fake(x, params, config):
    return x.new_empty((x.shape[0], engine_for(config).spec.node_dim))


def _scatter_args(src, index, dim, dim_size):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("mtmc_mpn.scatter_*: only dim=0 (the reference's call form) is implemented")
    if index.dim() != 1 or index.shape[0] != src.shape[0]:
        raise RuntimeError("mtmc_mpn.scatter_*: index must be 1-D with one entry per row of src")
    if not src.is_cuda or index.device != src.device:
        raise RuntimeError(f"mtmc_mpn.scatter_*: src and index must be on the same ROCm device (got {src.device} and "
                           f"{index.device}); there is no CPU path")
    if index.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("mtmc_mpn.scatter_*: index must be an integer tensor")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    return src.reshape(src.shape[0], -1).contiguous(), index.contiguous().long(), int(dim_size)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


@torch.library.impl("mtmc_mpn::scatter_add", "CUDA")
def _scatter_add(src, index, dim, dim_size):
    s2, idx, n = _scatter_args(src, index, dim, dim_size)
    lib = _lib.load()
    with torch.cuda.device(src.device):
        if not src.dtype.is_floating_point:        # utils.py:173-174: int64 in, int64 out
            s2 = s2.long()
            res = torch.empty((n, s2.shape[1]), dtype=torch.int64, device=src.device)
            _lib.check(lib.mtmc_scatter_add_i64(s2.data_ptr(), idx.data_ptr(), s2.shape[0], s2.shape[1], n,
                                                res.data_ptr(), _stream(src.device)))
            return res.reshape((n,) + tuple(src.shape[1:])).to(src.dtype)
        s2 = s2.float()
        res = torch.empty((n, s2.shape[1]), dtype=torch.float32, device=src.device)
        _lib.check(lib.mtmc_scatter_add(s2.data_ptr(), idx.data_ptr(), s2.shape[0], s2.shape[1], n, res.data_ptr(),
                                        _stream(src.device)))
    return res.reshape((n,) + tuple(src.shape[1:]))


@torch.library.impl("mtmc_mpn::scatter_mean", "CUDA")
def _scatter_mean(src, index, dim, dim_size):
    s2, idx, n = _scatter_args(src, index, dim, dim_size)
    s2 = s2.float()
    res = torch.empty((n, s2.shape[1]), dtype=torch.float32, device=src.device)
    cnt = torch.empty((max(n, 1),), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().mtmc_scatter_mean(s2.data_ptr(), idx.data_ptr(), s2.shape[0], s2.shape[1], n,
                                                 res.data_ptr(), cnt.data_ptr(), _stream(src.device)))
    return res.reshape((n,) + tuple(src.shape[1:]))


@torch.library.impl("mtmc_mpn::scatter_max", "CUDA")
def _scatter_max(src, index, dim, dim_size):
    s2, idx, n = _scatter_args(src, index, dim, dim_size)
    s2 = s2.float()
    res = torch.empty((n, s2.shape[1]), dtype=torch.float32, device=src.device)
    arg = torch.empty((n, s2.shape[1]), dtype=torch.int64, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().mtmc_scatter_max(s2.data_ptr(), idx.data_ptr(), s2.shape[0], s2.shape[1], n,
                                                res.data_ptr(), arg.data_ptr(), _stream(src.device)))
    shape = (n,) + tuple(src.shape[1:])
    return res.reshape(shape), arg.reshape(shape)


def _scatter_fake_shape(src, index, dim_size):
    if dim_size is None:
        raise RuntimeError("mtmc_mpn.scatter_*: dim_size is needed to trace the op")
    return (int(dim_size),) + tuple(src.shape[1:])


@torch.library.register_fake("mtmc_mpn::scatter_add")
def _scatter_add_fake(src, index, dim, dim_size):
    return src.new_empty(_scatter_fake_shape(src, index, dim_size),
                         dtype=src.dtype if not src.dtype.is_floating_point else torch.float32)


@torch.library.register_fake("mtmc_mpn::scatter_mean")
def _scatter_mean_fake(src, index, dim, dim_size):
    return src.new_empty(_scatter_fake_shape(src, index, dim_size), dtype=torch.float32)


@torch.library.register_fake("mtmc_mpn::scatter_max")
def _scatter_max_fake(src, index, dim, dim_size):
    shp = _scatter_fake_shape(src, index, dim_size)
    return src.new_empty(shp, dtype=torch.float32), src.new_empty(shp, dtype=torch.int64)
