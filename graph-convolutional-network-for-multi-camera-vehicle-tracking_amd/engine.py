"""Host side of one `MOTMPNet.forward`: turns the module's parameters and the caller's tensors into the
plain-pointer structs of include/mtmc_mpn.h and enqueues the whole forward on the current HIP stream
with ONE library call.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import types
from typing import List, Tuple

import torch

from . import _lib
from .config import MpnSpec


def _lin(mlp, spec_layer):
    return mlp.fc_layers[spec_layer.lin_slot], (mlp.fc_layers[spec_layer.bn_slot] if spec_layer.bn_slot is not None else None)


def _check_param(p: torch.Tensor, dev) -> int:
    if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
        raise RuntimeError("mtmc_mpn: parameters must be contiguous float32 tensors on the input's device "
                           f"(got {p.dtype} on {p.device})")
    if p.data_ptr() % 16:
        # the many-row GEMM kernels read weights with 16-byte vector loads and LDS-DMA; a Parameter that is a view at an odd
        # storage offset would otherwise fail on many-row graphs only (MTMC_E_ARG from the row-streaming kernel)
        raise RuntimeError("mtmc_mpn: parameters must be 16-byte aligned (a view at an odd storage offset? use .clone())")
    return p.data_ptr()


def layer_slots(spec: MpnSpec):
    """(struct slot, index or None, LayerSpec) for every layer, in mtmc_mpn_model order.  The flat parameter list every
    entry point takes follows it: weight, bias, then BatchNorm gamma, beta where the layer has one."""
    out = [("enc_node", i, l) for i, l in enumerate(spec.enc_node)]
    out += [("enc_edge", i, l) for i, l in enumerate(spec.enc_edge)]
    out += [("upd_edge", None, spec.upd_edge[0]), ("upd_node", None, spec.upd_node[0]), ("cls", None, spec.cls_edge[0])]
    return out


def module_layers(module):
    """(slot, Linear, BatchNorm or None, LayerSpec) of a MOTMPNet module tree, in layer_slots order."""
    m, s = module, module.spec
    out = []
    for i, layer in enumerate(s.enc_node):
        out.append((("enc_node", i),) + _lin(m.encoder.node_mlp, layer) + (layer,))
    for i, layer in enumerate(s.enc_edge):
        out.append((("enc_edge", i),) + _lin(m.encoder.edge_mlp, layer) + (layer,))
    out.append((("upd_edge", None),) + _lin(m.MPNet.edge_model.edge_mlp, s.upd_edge[0]) + (s.upd_edge[0],))
    out.append((("upd_node", None),) + _lin(m.MPNet.node_model.node_mlp, s.upd_node[0]) + (s.upd_node[0],))
    out.append((("cls", None),) + _lin(m.classifier.edge_mlp, s.cls_edge[0]) + (s.cls_edge[0],))
    return out


def ordered_params(module) -> List[torch.Tensor]:
    """The module's parameters as the flat list torch.ops.mtmc_mpn.* take (layer_slots order).  Read from the module tree on
    every call -- through the containers' `_modules` / `_parameters` dictionaries (what `nn.Module.__getattr__` ends up
    doing, without its fallback chain: 34 attribute reads cost 80 us per call the long way, a tenth of a training step's host
    time), so a replaced Parameter, Linear / BatchNorm1d or `fc_layers` container is still the one that is used."""
    m, s = module, module.spec
    mods = m._modules
    enc, mp, cls = mods["encoder"]._modules, mods["MPNet"]._modules, mods["classifier"]._modules
    groups = ((enc["node_mlp"], s.enc_node), (enc["edge_mlp"], s.enc_edge),
              (mp["edge_model"]._modules["edge_mlp"], s.upd_edge[:1]), (mp["node_model"]._modules["node_mlp"], s.upd_node[:1]),
              (cls["edge_mlp"], s.cls_edge[:1]))
    out = []

    def take(mod, name):
        t = mod._parameters.get(name)           # (a parametrized module keeps the attribute as a property: the long way then)
        out.append(t if t is not None else getattr(mod, name))
    for mlp, layers in groups:
        seq = mlp._modules["fc_layers"]._modules
        for layer in layers:
            lin = seq[str(layer.lin_slot)]
            take(lin, "weight")
            take(lin, "bias")
            if layer.bn_slot is not None:
                bn = seq[str(layer.bn_slot)]
                take(bn, "weight")
                take(bn, "bias")
    return out


class ForwardEngine:
    """Host side of the C ABI for one architecture (`MpnSpec`).  Constructed from a spec (parameters then come with
    every call, as torch.ops.mtmc_mpn.* pass them) or from a module (its own parameters are the default)."""

    def __init__(self, module_or_spec):
        if isinstance(module_or_spec, MpnSpec):
            self.module, self.spec = None, module_or_spec
        else:
            self.module, self.spec = module_or_spec, module_or_spec.spec
        self.lib = _lib.load()
        self._ws = {}
        self._slots = layer_slots(self.spec)
        self._ms_key, self._ms = None, None
        self._wc = {}                   # (device, stream) -> the weight-plane cache buffer of eval-mode calls on that stream
        self.weight_cache = True        # hand the library a weight-plane cache (torch_ops.NO_WEIGHT_CACHE / module.cache_weight_planes)
        self.flags = 0                  # MTMC_F_* for the calls this engine prepares (torch op argument)

    # -- parameters -> mtmc_mpn_model ------------------------------------------------------------
    def param_layers(self):
        """(struct slot, Linear, BatchNorm or None, LayerSpec) for every layer of the bound module."""
        if self.module is None:
            raise RuntimeError("mtmc_mpn: this engine was built from a spec; pass the parameter list explicitly")
        return module_layers(self.module)          # re-read per call: sub-modules / Parameters may have been replaced

    def params(self) -> List[torch.Tensor]:
        if self.module is None:
            raise RuntimeError("mtmc_mpn: no parameters given")
        return ordered_params(self.module)         # never cached: see MOTMPNet.forward

    def model_struct(self, dev, params=None) -> _lib.Model:
        params = self.params() if params is None else params
        # the struct only changes when a parameter's storage does: key it on the device pointers -- and on what _check_param
        # looks at, since an allocator can hand a recycled address to a tensor of another dtype or shape (e.g. after .half())
        key = (dev,) + tuple((t.data_ptr(), t.dtype, t.shape, t.stride()) for t in params)
        if key == self._ms_key:
            return _lib.Model.from_buffer_copy(self._ms)
        s = self.spec
        out = _lib.Model()
        if len(s.enc_node) > _lib.MAX_ENC_LAYERS:
            raise NotImplementedError("mtmc_mpn: node encoder deeper than 8 layers")
        it = iter(params)
        try:
            for slot, idx, layer in self._slots:
                dst = getattr(out, slot) if idx is None else getattr(out, slot)[idx]
                w, b = next(it), next(it)
                if tuple(w.shape) != (layer.out_dim, layer.in_dim) or tuple(b.shape) != (layer.out_dim,):
                    raise RuntimeError(f"mtmc_mpn: parameter shapes of {slot}[{idx}] do not match the configuration")
                dst.weight, dst.bias = _check_param(w, dev), _check_param(b, dev)
                if layer.bn_slot is not None:
                    dst.gamma, dst.beta = _check_param(next(it), dev), _check_param(next(it), dev)
                else:
                    dst.gamma = dst.beta = None
                dst.in_dim, dst.out_dim = layer.in_dim, layer.out_dim
        except StopIteration:
            raise RuntimeError("mtmc_mpn: parameter list shorter than the configuration needs") from None
        out.dropout_enc = float(s.enc_node[0].dropout_p or 0.0)
        out.dropout_upd_edge = float(s.upd_edge[0].dropout_p or 0.0)
        out.dropout_upd_node = float(s.upd_node[0].dropout_p or 0.0)
        out.n_enc_layers = len(s.enc_node)
        out.agg = _lib.AGG[s.agg]
        out.num_enc_steps, out.num_class_steps = s.num_enc_steps, s.num_class_steps
        out.reattach_nodes, out.reattach_edges = int(s.reattach_nodes), int(s.reattach_edges)
        self._ms_key, self._ms = key, out
        return _lib.Model.from_buffer_copy(out)

    def shape_model(self) -> _lib.Model:
        """The configuration's dimensions and flags with PLACEHOLDER parameter pointers (non-NULL, never dereferenced):
        for the host-only size queries (workspace bytes, gradient layout) that shape inference needs without tensors."""
        s = self.spec
        out = _lib.Model()
        for slot, idx, layer in self._slots:
            dst = getattr(out, slot) if idx is None else getattr(out, slot)[idx]
            dst.weight = dst.bias = 256
            dst.gamma = dst.beta = 256 if layer.bn_slot is not None else None
            dst.in_dim, dst.out_dim = layer.in_dim, layer.out_dim
        out.n_enc_layers = len(s.enc_node)
        out.agg = _lib.AGG[s.agg]
        out.num_enc_steps, out.num_class_steps = s.num_enc_steps, s.num_class_steps
        out.reattach_nodes, out.reattach_edges = int(s.reattach_nodes), int(s.reattach_edges)
        return out

    def workspace(self, model, n, e, dev, stream_ptr) -> torch.Tensor:
        need = self.lib.mtmc_mpn_workspace_bytes(C.byref(model), n, e)
        if need == 0:
            _lib.check(_lib.E_ARG)
        key = (dev, stream_ptr)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def weight_plane_cache(self, model, dev, stream_ptr) -> torch.Tensor:
        """mtmc_mpn_call::weight_cache: a zero-filled buffer per (device, stream) in which the library keeps the fp16 planes /
        row scales it derives from the node-encoder weights.  NOTHING here decides whether its content is still valid: every
        eval-mode forward re-reads the weights on the device and compares a 64-bit fingerprint per 8 weight rows with the one
        stored beside the planes (csrc/split_body.h), so in-place updates, writes through `param.data`, another module of the
        same configuration at recycled addresses and graph replays after a weight change all get the planes of the weights
        they run with (round 4 keyed the cache on (data_ptr, _version) on the host)."""
        key = (dev, stream_ptr)
        buf = self._wc.get(key)
        if buf is None:
            need = self.lib.mtmc_mpn_weight_cache_bytes(C.byref(model))
            if need == 0:
                _lib.check(_lib.E_ARG)
            buf = self._wc[key] = torch.zeros(need + 256, dtype=torch.uint8, device=dev)
        return buf

    # -- the call -----------------------------------------------------------------------------------
    def check_inputs(self, x, edge_index, edge_attr):
        if not (isinstance(x, torch.Tensor) and x.is_cuda and edge_index.is_cuda and edge_attr.is_cuda):
            raise RuntimeError("mtmc_mpn: data.x / edge_index / edge_attr must be on a ROCm GPU -- "
                               "this module has no CPU or PyTorch fallback path")
        if not (x.device == edge_index.device == edge_attr.device):
            raise RuntimeError("mtmc_mpn: all inputs must be on the same device")
        s = self.spec
        if x.dim() != 2 or x.shape[1] != s.enc_node[0].in_dim:
            raise RuntimeError(f"mtmc_mpn: data.x must be [N, {s.enc_node[0].in_dim}], got {tuple(x.shape)}")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError(f"mtmc_mpn: data.edge_index must be [2, E], got {tuple(edge_index.shape)}")
        e = edge_index.shape[1]
        if edge_attr.dim() != 2 or tuple(edge_attr.shape) != (e, s.enc_edge[0].in_dim):
            raise RuntimeError(f"mtmc_mpn: data.edge_attr must be [{e}, {s.enc_edge[0].in_dim}], got {tuple(edge_attr.shape)}")
        if x.dtype != torch.float32 or edge_attr.dtype != torch.float32:
            raise RuntimeError("mtmc_mpn: x and edge_attr must be float32")
        if edge_index.dtype != torch.int64:
            raise RuntimeError("mtmc_mpn: edge_index must be int64")

    def prepare(self, x, edge_index, edge_attr, training=False, n_edges_total=None, node_range=None, tape=False,
                seed=0, row_range=None, params=None, tape_ws=None):
        """Validate, allocate outputs/workspace and fill the two C structs of one call.
        `tape=True`: training-mode layout in a fresh workspace that the backward will read (kept by autograd).
        `row_range=(lo, hi)`: multi-GPU, row-complete edge shard -- project / take node statistics of these rows only
        (mtmc_mpn_call::row_lo / row_hi)."""
        self.check_inputs(x, edge_index, edge_attr)
        s, dev = self.spec, x.device
        e = edge_index.shape[1]
        node_lo, node_hi = (0, x.shape[0]) if node_range is None else node_range[:2]
        n = x.shape[0] if node_range is None else node_range[2]
        # the kernels index x by (node - node_lo): a slice of the wrong height would be read out of bounds on the device
        if not (0 <= node_lo <= node_hi <= n) or x.shape[0] != node_hi - node_lo:
            raise RuntimeError(f"mtmc_mpn: node_range {(node_lo, node_hi, n)} does not describe the {x.shape[0]} rows of x")
        if row_range is not None and not (0 <= int(row_range[0]) <= int(row_range[1]) <= n):
            raise RuntimeError(f"mtmc_mpn: row_range {tuple(row_range)} is not inside [0, {n}]")
        if x.stride(1) != 1 or x.stride(0) % 4 != 0:
            x = x.contiguous()
        if not edge_attr.is_contiguous():
            edge_attr = edge_attr.contiguous()
        if e > 0 and edge_index.stride(1) < 1:
            edge_index = edge_index.contiguous()
        n_out = min(s.num_class_steps, s.num_enc_steps) if s.num_enc_steps > 0 else 1
        n_cls = s.cls_edge[0].out_dim
        logits = torch.empty((n_out, e, n_cls), dtype=torch.float32, device=dev)
        h = torch.empty((n, s.node_dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
        model = self.model_struct(dev, params)
        if tape_ws is not None:                        # backward: the tape the forward op returned
            ws = tape_ws
        elif tape:
            need = self.lib.mtmc_mpn_train_workspace_bytes(C.byref(model), n, e)
            if need == 0:
                _lib.check(_lib.E_ARG)
            ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        else:
            ws = self.workspace(model, n, e, dev, stream)
        call = _lib.Call()
        call.x, call.x_row_stride = x.data_ptr(), x.stride(0)
        call.row = edge_index[0].data_ptr() if e > 0 else None
        call.col = edge_index[1].data_ptr() if e > 0 else None
        call.idx_stride = edge_index.stride(1) if e > 0 else 1
        call.edge_attr = edge_attr.data_ptr() if e > 0 else ws.data_ptr()
        call.n_nodes, call.n_edges = n, e
        call.n_edges_total = e if n_edges_total is None else n_edges_total
        call.node_lo, call.node_hi = node_lo, node_hi
        if row_range is not None:                      # (k, k) with k > 0 is the empty range; (0, 0) means "all"
            rlo, rhi = int(row_range[0]), int(row_range[1])
            call.row_lo, call.row_hi = (rlo, rhi) if rhi > rlo else (max(rlo, 1), max(rlo, 1))
        call.logits = logits.data_ptr() if logits.numel() else ws.data_ptr()
        call.h_out = h.data_ptr()
        call.workspace, call.workspace_bytes = ws.data_ptr(), ws.numel()
        call.training, call.seed = (1 if tape else 0), int(seed) & 0xFFFFFFFFFFFFFFFF
        call.flags = int(self.flags) | (_lib.F_DETERMINISTIC if getattr(self.module, "deterministic", False) else 0)
        keep = (x, edge_index, edge_attr)        # the structs hold raw pointers: keep the tensors alive
        if self.weight_cache and tape_ws is None:      # (training forwards too: every chunk whose weights the step changed is split again)
            wc = self.weight_plane_cache(model, dev, stream)
            call.weight_cache, call.weight_cache_bytes = wc.data_ptr(), wc.numel()
            keep = keep + (wc,)
        call.stream = stream
        return types.SimpleNamespace(model=model, call=call, ws=ws, logits=logits, h=h, n_out=n_out, n=n, e=e,
                                     dev=dev, keep=keep)

    def plan(self, n_nodes, n_edges, n_edges_total=None, node_range=None, row_range=None, training=False, flags=0,
             params=None, device=None, weight_cache=True) -> types.SimpleNamespace:
        """Which kernels a call of these sizes would run (mtmc_mpn_plan_call: host-only, nothing is launched and no GPU
        is needed) -- lets a multi-GPU host, or a test, check that a shard takes the kernels the whole graph would."""
        params = self.params() if params is None else params
        model = self.model_struct(params[0].device if device is None else device, params)
        call = _lib.Call()
        call.n_nodes, call.n_edges = int(n_nodes), int(n_edges)
        call.n_edges_total = int(n_edges if n_edges_total is None else n_edges_total)
        call.node_lo, call.node_hi = (0, int(n_nodes)) if node_range is None else (int(node_range[0]), int(node_range[1]))
        if row_range is not None:
            rlo, rhi = int(row_range[0]), int(row_range[1])
            call.row_lo, call.row_hi = (rlo, rhi) if rhi > rlo else (max(rlo, 1), max(rlo, 1))
        call.training, call.flags = int(bool(training)), int(flags)
        if weight_cache:                               # (tested for NULL by the query, never read)
            call.weight_cache, call.weight_cache_bytes = 256, 1 << 62
        out = _lib.Plan()
        _lib.check(self.lib.mtmc_mpn_plan_call(C.byref(model), C.byref(call), C.byref(out)))
        n_layers = len(self.spec.enc_node)
        return types.SimpleNamespace(enc_kernel=list(out.enc_kernel)[:n_layers], enc_split_k=list(out.enc_split_k)[:n_layers],
                                     edges_per_thread=out.edges_per_thread, lazy_edges=bool(out.lazy_edges),
                                     pass_c=out.pass_c, avg_degree=out.avg_degree, pass_a_col_blocks=out.pass_a_col_blocks,
                                     layer0_panels=out.layer0_panels, enc2_passenger=bool(out.enc2_passenger),
                                     node_stat_folded=bool(out.node_stat_folded))

    def phase_list(self):
        """(phase, arg) pairs of one forward, in order (what mtmc_mpn_forward runs internally)."""
        s = self.spec
        seq = [(_lib.PH_BEGIN, 0), (_lib.PH_EDGE_ENC, 0)]
        for l in range(len(s.enc_node)):
            seq += [(_lib.PH_NODE_ENC, l), (_lib.PH_NODE_COMBINE, l)]
        seq += [(_lib.PH_NODE_H0, 0)]
        for r in range(s.num_enc_steps):
            seq += [(_lib.PH_ROUND_PROJ, r), (_lib.PH_ROUND_A, r), (_lib.PH_ROUND_B, r), (_lib.PH_ROUND_STAT, r),
                    (_lib.PH_ROUND_C, r)]
        return seq + [(_lib.PH_END, 0)]

    # -- phase backend interface used by mtmc_mpn.distributed ------------------------------------------
    def run_phase(self, prep, ph, arg):
        with torch.cuda.device(prep.dev):
            _lib.check(self.lib.mtmc_mpn_run_phase(C.byref(prep.model), C.byref(prep.call), ph, arg))

    def run_phase_list(self, prep, pairs):
        """Several consecutive phases in one library call (mtmc_mpn_run_phases): what runs between two collectives."""
        flat = (C.c_int32 * (2 * len(pairs)))(*[v for pair in pairs for v in pair])
        with torch.cuda.device(prep.dev):
            _lib.check(self.lib.mtmc_mpn_run_phases(C.byref(prep.model), C.byref(prep.call), flat, len(pairs)))

    def set_flags(self, prep, flags):
        prep.call.flags = int(flags)

    def region(self, prep, name, idx=0) -> torch.Tensor:
        """A tensor aliasing one exchanged region of the workspace (include/mtmc_mpn.h, mtmc_ws_layout)."""
        lay = getattr(prep, "_layout", None)
        if lay is None:
            lay = prep._layout = self.layout(prep)
        ws, n, s = prep.ws, prep.n, self.spec
        R = _lib.STAT_REPLICAS

        def f64(off, count):
            return ws[off:off + 8 * count].view(torch.float64)

        if name == "stat_attr":
            return f64(lay.stat_attr_off, R * _lib.ATTR_STRIDE)
        if name == "stat_enc2":
            return f64(lay.stat_enc2_off, R * _lib.ENC2_STRIDE)
        if name == "stat_enc_node":
            return f64(lay.stat_enc_layer_off[idx], 2 * s.enc_node[idx].out_dim)
        if name == "enc_merged":               # the edge branch's block + encoder layer idx's (idx 0: stat_attr, 1: stat_enc2),
            head = lay.stat_attr_off if idx == 0 else lay.stat_enc2_off      # adjacent in the workspace: ONE message
            end = lay.stat_enc_layer_off[idx] + 16 * s.enc_node[idx].out_dim
            return (f64(head, (end - head) // 8),)
        if name == "round_m_z2":               # adjacent in the round block: one contiguous message
            base = lay.stat_round_off + 8 * idx * _lib.ROUND_BLOCK + 8 * R * _lib.Z1_STRIDE
            return (f64(base, R * (_lib.M_STRIDE + _lib.Z2_STRIDE)),)
        if name in ("round_z1", "round_m", "round_z2"):
            base = lay.stat_round_off + 8 * idx * _lib.ROUND_BLOCK
            if name == "round_z1":
                return f64(base, R * _lib.Z1_STRIDE)
            if name == "round_m":
                return f64(base + 8 * R * _lib.Z1_STRIDE, R * _lib.M_STRIDE)
            return f64(base + 8 * R * (_lib.Z1_STRIDE + _lib.M_STRIDE), R * _lib.Z2_STRIDE)
        if name == "deg":
            return ws[lay.deg_off:lay.deg_off + 4 * n].view(torch.int32)
        if name == "deg_global":
            return ws[lay.deg_global_off:lay.deg_global_off + 4 * n].view(torch.int32)
        if name == "h0":
            return ws[lay.h0_off:lay.h0_off + 128 * n].view(torch.float32).view(n, 32)
        if name == "Pc":                       # the round's column projections [N][4] (second half of P)
            off = lay.P_off + 16 * n
            return ws[off:off + 16 * n].view(torch.float32).view(n, 4)
        if name == "agg":                      # where round idx aggregates (see mtmc_ws_layout.h_acc_off)
            if idx == s.num_enc_steps - 1 and s.agg != "mean":
                return prep.h
            off = lay.h_acc_off[idx & 1]
            return ws[off:off + 128 * n].view(torch.float32).view(n, 32)
        raise KeyError(name)

    def outputs(self, prep):
        return [prep.logits[i] for i in range(prep.n_out)], prep.h

    def layout(self, prep) -> _lib.WsLayout:
        lay = _lib.WsLayout()
        _lib.check(self.lib.mtmc_mpn_workspace_layout(C.byref(prep.model), prep.n, prep.e, C.byref(lay)))
        return lay

    def run_phases(self, prep, after_phase=None, before_phase=None):
        """Phase-by-phase forward (same kernels as __call__): the multi-GPU host interleaves collectives
        through `after_phase(phase, arg)`, bench.py brackets phases with events."""
        with torch.cuda.device(prep.dev):
            for ph, arg in self.phase_list():
                if before_phase is not None:
                    before_phase(ph, arg)
                _lib.check(self.lib.mtmc_mpn_run_phase(C.byref(prep.model), C.byref(prep.call), ph, arg))
                if after_phase is not None:
                    after_phase(ph, arg)
        return [prep.logits[i] for i in range(prep.n_out)], prep.h

    def __call__(self, x, edge_index, edge_attr, training=False) -> Tuple[List[torch.Tensor], torch.Tensor]:
        """The bound module's forward, through the registered op (torch.ops.mtmc_mpn.mp_forward)."""
        if self.module is None:
            raise RuntimeError("mtmc_mpn: engine built from a spec has no module to call")
        out, h = self.module(types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr))
        return out["classified_edges"], h
