"""Model-shape contract of the message-passing network.

`DEFAULT_GRAPH_NET_PARAMS` restates the GRAPH_NET_PARAMS block the reference ships
(config/config_training.yaml:68-111); `resolve()` turns such a dict (plus the CNN arch key the
callers pass, main.py:96, main_training.py:211) into the flat `MpnSpec` the HIP path is
specialised for, validating everything the kernels assume.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

DEFAULT_ARCH = "resnet101"

DEFAULT_GRAPH_NET_PARAMS: Dict = {
    "node_agg_fn": "sum",
    "num_enc_steps": 1,
    "num_class_steps": 1,
    "reattach_initial_nodes": False,
    "reattach_initial_edges": False,
    "encoder_feats_dict": {
        "edges": {"edge_in_dim": 2, "edge_fc_dims": [4], "edge_out_dim": 4},
        "nodes": {
            "resnet101": {
                "node_in_dim": 2048,
                "node_fc_dims": [1024, 512, 128],
                "node_out_dim": 32,
                "dropout_p": 0.1,
                "use_batchnorm": True,
            }
        },
    },
    "edge_model_feats_dict": {"fc_dims": [4], "dropout_p": 0.1, "use_batchnorm": True},
    "node_model_feats_dict": {"fc_dims": [32], "dropout_p": 0.1, "use_batchnorm": True},
    "classifier_feats_dict": {
        "edge_in_dim": 4,
        "edge_fc_dims": [],
        "edge_out_dim": 2,
        "dropout_p": 0,
        "use_batchnorm": False,
        "is_classifier": True,
    },
}


def default_params(**overrides) -> Dict:
    """Deep copy of the shipped config with top-level keys overridden (e.g. num_enc_steps=3)."""
    p = copy.deepcopy(DEFAULT_GRAPH_NET_PARAMS)
    p.update(overrides)
    return p


@dataclass
class LayerSpec:
    """One `Linear [-> BatchNorm1d] [-> ReLU] [-> Dropout]` group of a reference MLP and the
    nn.Sequential slots it occupies (reference models/mlp.py:11-27)."""
    in_dim: int
    out_dim: int
    lin_slot: int
    bn_slot: Optional[int]
    relu: bool
    dropout_p: Optional[float]


def plan_mlp(input_dim: int, fc_dims, dropout_p, use_batchnorm, is_classifier=False) -> List[LayerSpec]:
    assert isinstance(fc_dims, (list, tuple)), \
        "fc_dims must be either a list or a tuple, but got {}".format(type(fc_dims))
    layers, slot = [], 0
    for dim in fc_dims:
        lin, bn, relu, drop = slot, None, False, None
        slot += 1
        if not is_classifier and dim != 1:
            if use_batchnorm:
                bn, slot = slot, slot + 1
            relu, slot = True, slot + 1
            if dropout_p is not None:
                drop, slot = float(dropout_p), slot + 1
        layers.append(LayerSpec(input_dim, int(dim), lin, bn, relu, drop))
        if not is_classifier:      # the reference's classifier branch never advances input_dim (models/mlp.py:25-27)
            input_dim = int(dim)
    return layers


@dataclass
class MpnSpec:
    """Flat description of one MOTMPNet instance (everything reference models/mpn.py:154-248 derives)."""
    enc_node: List[LayerSpec]
    enc_edge: List[LayerSpec]
    cls_edge: List[LayerSpec]
    upd_edge: List[LayerSpec]
    upd_node: List[LayerSpec]
    agg: str
    num_enc_steps: int
    num_class_steps: int
    reattach_nodes: bool
    reattach_edges: bool
    node_dim: int = field(init=False)
    edge_dim: int = field(init=False)

    def __post_init__(self):
        self.node_dim = self.enc_node[-1].out_dim
        self.edge_dim = self.enc_edge[-1].out_dim


def resolve(model_params: Dict, arch: Optional[str]) -> MpnSpec:
    enc_e = model_params["encoder_feats_dict"]["edges"]
    enc_n = model_params["encoder_feats_dict"]["nodes"][arch]
    cls = model_params["classifier_feats_dict"]
    em = model_params["edge_model_feats_dict"]
    nm = model_params["node_model_feats_dict"]
    agg = model_params["node_agg_fn"]
    assert agg.lower() in ("mean", "max", "sum"), "node_agg_fn can only be 'max', 'mean' or 'sum'."
    re_n = bool(model_params["reattach_initial_nodes"])
    re_e = bool(model_params["reattach_initial_edges"])
    nf, ef = (2 if re_n else 1), (2 if re_e else 1)
    h, he = enc_n["node_out_dim"], enc_e["edge_out_dim"]
    # the reference merges the node dict into the edge dict (models/mpn.py:169), so the edge
    # encoder inherits the node encoder's dropout_p / use_batchnorm
    return MpnSpec(
        enc_node=plan_mlp(enc_n["node_in_dim"], list(enc_n["node_fc_dims"]) + [h],
                          enc_n["dropout_p"], enc_n["use_batchnorm"]),
        enc_edge=plan_mlp(enc_e["edge_in_dim"], list(enc_e["edge_fc_dims"]) + [he],
                          enc_n["dropout_p"], enc_n["use_batchnorm"]),
        cls_edge=plan_mlp(cls["edge_in_dim"], list(cls["edge_fc_dims"]) + [cls["edge_out_dim"]],
                          cls["dropout_p"], cls["use_batchnorm"], cls.get("is_classifier", False)),
        upd_edge=plan_mlp(nf * 2 * h + ef * he, em["fc_dims"], em["dropout_p"], em["use_batchnorm"]),
        upd_node=plan_mlp(nf * h + he, nm["fc_dims"], nm["dropout_p"], nm["use_batchnorm"]),
        agg=agg.lower(), num_enc_steps=int(model_params["num_enc_steps"]),
        num_class_steps=int(model_params["num_class_steps"]),
        reattach_nodes=re_n, reattach_edges=re_e)


# ---- what the gfx950 kernels are specialised for ------------------------------------------
NODE_DIM = 32      # H : node state width the round kernels are compiled for
EDGE_DIM = 4       # He: edge state width
MAX_EDGE_IN = 2    # raw edge_attr width
MAX_CLS_DIM = 4


def check_supported(spec: MpnSpec) -> Tuple[bool, str]:
    """The HIP path covers the shipped architecture family (config_training.yaml:68-111 and the
    variants main_training.py:94-99 / the reattach and aggregation flags produce).  Anything
    else is rejected loudly rather than computed some other way."""
    def hidden(layers, n=None):
        return all(l.bn_slot is not None and l.relu for l in layers) and (n is None or len(layers) == n)
    if spec.node_dim != NODE_DIM or spec.edge_dim != EDGE_DIM:
        return False, f"node/edge state widths must be {NODE_DIM}/{EDGE_DIM}, got {spec.node_dim}/{spec.edge_dim}"
    if not hidden(spec.enc_node) or len(spec.enc_node) < 1:
        return False, "node encoder layers must all be Linear+BatchNorm+ReLU"
    if not hidden(spec.enc_edge, 2) or spec.enc_edge[0].in_dim not in (1, 2) or spec.enc_edge[0].out_dim != EDGE_DIM:
        return False, "edge encoder must be in(1|2) -> 4 -> 4 with BatchNorm"
    if not hidden(spec.upd_edge, 1) or spec.upd_edge[0].out_dim != EDGE_DIM:
        return False, "edge update MLP must be one hidden layer of width 4 with BatchNorm"
    if not hidden(spec.upd_node, 1) or spec.upd_node[0].out_dim != NODE_DIM:
        return False, "node update MLP must be one hidden layer of width 32 with BatchNorm"
    if len(spec.cls_edge) != 1 or spec.cls_edge[0].bn_slot is not None or spec.cls_edge[0].relu \
            or spec.cls_edge[0].in_dim != EDGE_DIM or not (1 <= spec.cls_edge[0].out_dim <= MAX_CLS_DIM):
        # (a 2-layer classifier, main_training.py:94-99, cannot run in the reference either:
        #  its second Linear is built with the first one's input width, models/mlp.py:25-27)
        return False, "classifier must be one bare Linear layer on the 4-d edge state"
    if spec.num_enc_steps < 0 or spec.num_class_steps < 0:
        return False, "negative step counts"
    return True, ""
