"""Drop-in `MOTMPNet` for the reference's message-passing network, backed by gfx950 HIP kernels.

Host-side mirror of the reference interface (reference models/mpn.py:144-299, models/mlp.py:4-33):
same constructor signature, same module tree -- hence the same 34 `state_dict` keys and shapes,
so `utils.load_pretrained_weights` / `load_state_dict(strict=True)` work unchanged -- same
`forward(data) -> ({'classified_edges': [Tensor[E,2], ...]}, Tensor[N,32])` contract.

The sub-modules only *hold parameters*; `MOTMPNet.forward` hands the whole forward to the
C-ABI library (csrc/, include/mtmc_mpn.h) in one call.  There is no PyTorch or CPU fallback:
without the built extension, or for tensors that are not on a ROCm device, forward raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib as _lib_flags
from . import config as _config
from .config import LayerSpec, MpnSpec


class MLP(nn.Module):
    """Parameter container with the slot layout of the reference MLP (models/mlp.py:11-30):
    `fc_layers.{slot}` is a Linear, BatchNorm1d(track_running_stats=False), ReLU or Dropout."""

    def __init__(self, input_dim, fc_dims, dropout_p=0.4, use_batchnorm=False, is_classifier=False):
        super().__init__()
        self.layers: List[LayerSpec] = _config.plan_mlp(input_dim, fc_dims, dropout_p, use_batchnorm, is_classifier)
        seq: List[nn.Module] = []
        for spec in self.layers:
            seq.append(nn.Linear(spec.in_dim, spec.out_dim))
            if spec.bn_slot is not None:
                seq.append(nn.BatchNorm1d(spec.out_dim, track_running_stats=False))
            if spec.relu:
                seq.append(nn.ReLU(inplace=True))
            if spec.dropout_p is not None:
                seq.append(nn.Dropout(p=spec.dropout_p))
        self.fc_layers = nn.Sequential(*seq)

    def forward(self, input):
        from . import ops
        return ops.mlp_forward(self, input)


class MLPGraphIndependent(nn.Module):
    """Encoder / classifier pair of independent node and edge MLPs (reference models/mpn.py:103-142)."""

    def __init__(self, edge_in_dim=None, node_in_dim=None, edge_out_dim=None, node_out_dim=None,
                 node_fc_dims=None, edge_fc_dims=None, dropout_p=None, use_batchnorm=None, is_classifier=False):
        super().__init__()
        # node MLP first, then edge MLP: parameter creation order fixes the RNG stream, so a
        # seeded construction yields the reference's initial weights bit for bit
        self.node_mlp = None if node_in_dim is None else MLP(
            node_in_dim, list(node_fc_dims) + [node_out_dim], dropout_p, use_batchnorm, is_classifier)
        self.edge_mlp = None if edge_in_dim is None else MLP(
            edge_in_dim, list(edge_fc_dims) + [edge_out_dim], dropout_p, use_batchnorm, is_classifier)

    def forward(self, edge_feats=None, nodes_feats=None):
        out_nodes = self.node_mlp(nodes_feats) if (self.node_mlp is not None and nodes_feats is not None) else nodes_feats
        out_edges = self.edge_mlp(edge_feats) if (self.edge_mlp is not None and edge_feats is not None) else edge_feats
        return out_edges, out_nodes


class EdgeModel(nn.Module):
    """Holds the edge-update MLP (reference models/mpn.py:59-69)."""

    def __init__(self, edge_mlp):
        super().__init__()
        self.edge_mlp = edge_mlp


class NodeModel(nn.Module):
    """Holds the node-update MLP and the aggregation name (reference models/mpn.py:71-101)."""

    def __init__(self, node_mlp, node_agg_fn: str):
        super().__init__()
        self.node_mlp = node_mlp
        self.node_agg_fn = node_agg_fn


class MetaLayer(nn.Module):
    """One message-passing round = edge update then node update (reference models/mpn.py:10-57)."""

    def __init__(self, edge_model=None, node_model=None):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model

    def __repr__(self):
        return "{}(edge_model={}, node_model={})".format(self.__class__.__name__, self.edge_model, self.node_model)


class MOTMPNet(nn.Module):
    """`MOTMPNet(model_params, bb_encoder=None, arch=...)`, call sites main.py:96, main_training.py:211."""

    def __init__(self, model_params: Dict, bb_encoder=None, arch: Optional[str] = None):
        super().__init__()
        self.node_cnn = bb_encoder          # stored, never used (reference models/mpn.py:163)
        self.model_params = model_params
        self.arch = arch
        self.spec: MpnSpec = _config.resolve(model_params, arch)
        ok, why = _config.check_supported(self.spec)
        if not ok:
            raise NotImplementedError("mtmc_mpn HIP path does not cover this GRAPH_NET_PARAMS: " + why)

        enc_e = model_params["encoder_feats_dict"]["edges"]
        enc_n = model_params["encoder_feats_dict"]["nodes"][arch]
        # same in-place merge as the reference (models/mpn.py:169): callers may re-read the dict
        enc_e.update(enc_n)
        self.encoder = MLPGraphIndependent(**enc_e)
        self.classifier = MLPGraphIndependent(**model_params["classifier_feats_dict"])
        em, nm = model_params["edge_model_feats_dict"], model_params["node_model_feats_dict"]
        s = self.spec
        self.MPNet = MetaLayer(
            edge_model=EdgeModel(MLP(s.upd_edge[0].in_dim, em["fc_dims"], em["dropout_p"], em["use_batchnorm"])),
            node_model=NodeModel(MLP(s.upd_node[0].in_dim, nm["fc_dims"], nm["dropout_p"], nm["use_batchnorm"]),
                                 s.agg))
        self.reattach_initial_nodes = s.reattach_nodes
        self.reattach_initial_edges = s.reattach_edges
        self.num_enc_steps = s.num_enc_steps
        self.num_class_steps = s.num_class_steps
        self.check_indices = False          # True: synchronise and raise IndexError on out-of-range edge_index
        self.deterministic = False          # True: order-independent sum/mean aggregation on row-sorted edge lists
        # Eval mode: the fp16 operand planes of the node-encoder WEIGHTS are kept from one forward to the next in a buffer the
        # library verifies against the weights' content on the device on every call (64-bit fingerprints per 8 weight rows,
        # engine.weight_plane_cache): any way of changing a weight -- optimizer steps, `param.data` writes, a new module at the
        # old addresses -- is seen.  False: no cache (few-row graphs then run the split-K kernels of rounds 1-4).
        self.cache_weight_planes = True
        # Training: None -- every forward draws its Dropout seed from torch's CPU generator (follows torch.manual_seed), a host
        # value.  An int64 [1] tensor on the model's device -- the seed is read from it ON THE DEVICE and it moves on by one per
        # forward (MTMC_F_SEED_ON_DEVICE): what a HIP graph of a whole training step needs (capture_training_step sets it).
        self.device_seed = None
        self._engine = None

    # -- the hot path -------------------------------------------------------------------
    def capture(self, data):
        """Record one eval-mode forward on `data` into a HIP graph and return a callable that replays it:
        `replay = model.capture(data); outputs, h = replay()`.  The tensors of `data` are captured by reference
        (write new values of the same shapes into them between replays); the returned tensors are overwritten by
        every replay.  For streams of same-sized graphs on hosts where ~25 kernel launches per forward cost more
        CPU time than the forward takes on the GPU; weights may change between replays (they are read in place)."""
        if self.training:
            raise RuntimeError("mtmc_mpn: capture() is for eval-mode inference")
        from . import torch_ops
        dev = data.x.device
        with torch.no_grad():
            self.forward(data)                            # engines exist from here on
            op_engine = torch_ops.engine_for(self._config_key)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                 # warm-up outside the capture: lazy library setup, and the workspace +
                self.forward(data)                        # weight-plane cache of this stream exist (and are verified) before it
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):    # captured on the SAME stream: nothing of the engine's is allocated inside
                out = self.forward(data)
            torch.cuda.current_stream(dev).wait_stream(side)
            # the buffers the captured kernels use must live exactly as long as the graph: take them out of the engine's
            # per-stream dictionaries (nothing else uses this stream; a later, larger forward would otherwise replace them)
            key = (dev, side.cuda_stream)
            held = [buf for buf in (op_engine._ws.pop(key, None), op_engine._wc.pop(key, None)) if buf is not None]
            held.append(side)

        def replay():
            graph.replay()
            return out
        replay.graph, replay.workspaces = graph, held
        return replay

    def capture_training_step(self, data, loss_fn, optimizer, warmup=3):
        """Record ONE training step -- `outputs, h = model(data); loss = loss_fn(outputs, h); loss.backward(); optimizer.step()`
        (reference train.py:356-424) -- into a HIP graph and return `replay() -> loss` (the loss tensor is overwritten by every
        replay; gradients are written afresh by every replay, as after `zero_grad(set_to_none=True)`).  A config-3 step is ~65
        kernel launches for 0.6 ms of GPU time: issued one by one the host sets the pace (0.65-0.9 ms from box to box), replayed
        it is the GPU's time.  Dropout masks change from replay to replay: the seed lives in `self.device_seed`, a device counter
        the forward reads and advances on the stream (drawn once, here, from torch's generator if it is None).
        Nothing may still hold an EARLIER step's autograd graph (its loss, its outputs): delete those references first.
        `warmup` REAL steps run first on the capture stream (lazy setup; the engine's buffers for that stream exist before the
        capture).  The tensors of `data` (and whatever `loss_fn` closes over) are captured by reference: write the next graph's
        values of the same shapes into them between replays.  `optimizer` must be capture-safe (e.g. SGD(fused=True)); its
        hyper-parameters are baked into the graph as plain numbers (a tensor `lr` is read at replay time)."""
        if not self.training:
            raise RuntimeError("mtmc_mpn: capture_training_step() is for .train() mode")
        dev = data.x.device
        if self.device_seed is None:
            self.device_seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(dev)    # follows torch.manual_seed
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

        def step():
            outputs, h = self(data)
            loss = loss_fn(outputs, h)
            loss.backward()
            optimizer.step()
            return loss
        import gc
        import warnings
        gc.collect()
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)                           # (the warning looked for below is a warn-once one)
        try:
            with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                for _ in range(max(int(warmup), 1)):
                    optimizer.zero_grad(set_to_none=True)
                    step()
            side.synchronize()
        finally:
            torch.set_warn_always(warn_always)
        for w in caught:
            if "AccumulateGrad node's stream does not match" in str(w.message):
                # gradient-accumulation nodes of an EARLIER step on another stream are still alive (something still holds that
                # step's autograd graph: its loss or outputs).  They would run on that stream during the capture and break it
                # (on this ROCm build: a crash, not an exception) -- refuse before trying.
                raise RuntimeError("mtmc_mpn.capture_training_step: an earlier step's autograd graph is still alive (its loss or "
                                   "outputs are referenced somewhere): delete those references (`del loss, outputs`) and call again")
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        optimizer.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            loss = step()
        torch.cuda.current_stream(dev).wait_stream(side)
        from . import torch_ops
        eng = torch_ops.engine_for(self._config_key)
        key = (dev, side.cuda_stream)
        held = [buf for buf in (eng._ws.pop(key, None), eng._wc.pop(key, None)) if buf is not None] + [side]

        def replay():
            graph.replay()
            return loss
        replay.graph, replay.workspaces = graph, held
        return replay

    def forward(self, data):
        """`outputs, latent_node_feats = mpn_model(data)` (reference inference.py:469, train.py:356): one call of the
        registered op `torch.ops.mtmc_mpn.mp_forward`; under grad mode / `.train()` it records the tape its autograd
        formula (`mp_backward`) consumes."""
        from . import engine, torch_ops
        if self._engine is None:
            self._engine = engine.ForwardEngine(self)
            self._config_key = torch_ops.config_key(self.model_params, self.arch)
        x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr      # (shapes / dtypes: engine.prepare)
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("mtmc_mpn: data.x / edge_index / edge_attr must be on a ROCm GPU -- "
                               "this module has no CPU or PyTorch fallback path")
        # read from the module tree on EVERY call (34 attribute reads): a Parameter object replaced after the first forward
        # (load_state_dict(assign=True), to_empty, parametrizations, `layer.weight = nn.Parameter(...)`) must be the one
        # that is used and that receives the gradient
        params = engine.ordered_params(self)
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or edge_attr.requires_grad or any(p.requires_grad for p in params))
        tape = bool(needs_grad or self.training)
        flags = (_lib_flags.F_DETERMINISTIC if self.deterministic else 0) | \
            (torch_ops.CHECK_INDICES if self.check_indices else 0) | \
            (0 if self.cache_weight_planes else torch_ops.NO_WEIGHT_CACHE)
        if self.training and self.device_seed is not None:
            ds = self.device_seed
            if not (isinstance(ds, torch.Tensor) and ds.dtype == torch.int64 and ds.numel() == 1 and ds.device == x.device):
                raise RuntimeError("mtmc_mpn: device_seed must be an int64 tensor of one element on the device of data.x")
            seed, flags = ds.data_ptr(), flags | _lib_flags.F_SEED_ON_DEVICE
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0    # follows torch.manual_seed
        if tape and edge_index.is_cuda and edge_attr.is_cuda and not torch.compiler.is_compiling():
            # grad mode: the op's own forward / backward functions without the dispatcher around them (torch_ops._MpForwardLean)
            steps, h = torch_ops.mp_forward_lean(x, edge_index, edge_attr, params, self._config_key, self.training, seed, flags)
            return {"classified_edges": steps}, h
        else:
            logits, h, _ = torch.ops.mtmc_mpn.mp_forward(x, edge_index, edge_attr, params, self._config_key,
                                                         self.training, seed, flags, tape)
        return {"classified_edges": list(logits.unbind(0))}, h
