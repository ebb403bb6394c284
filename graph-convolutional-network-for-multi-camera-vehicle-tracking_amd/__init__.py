"""mtmc_mpn -- MI355X (gfx950) native message-passing network for multi-camera tracklet association.

Import it as `mtmc_mpn` (the alias package at the repo root points here; this directory's
name is fixed by the build contract and is not a valid Python identifier).

Public surface = the reference's interface for the hot path (reference models/mpn.py, models/mlp.py):
`MOTMPNet`, `MetaLayer`, `EdgeModel`, `NodeModel`, `MLPGraphIndependent`, `MLP`, and the
`scatter_add / scatter_mean / scatter_max` functional surface of the third-party op it used.
"""
from .config import DEFAULT_ARCH, DEFAULT_GRAPH_NET_PARAMS, default_params  # noqa: F401
from .graph_build import build_graph  # noqa: F401
from .postprocess import postprocess  # noqa: F401
from . import ops  # noqa: F401
from .ops import cross_entropy, cross_entropy_steps  # noqa: F401
from .feature_store import FeatureStore  # noqa: F401
from .modules import (MLP, EdgeModel, MetaLayer, MLPGraphIndependent, MOTMPNet,  # noqa: F401
                      NodeModel)

__all__ = ["MOTMPNet", "MetaLayer", "EdgeModel", "NodeModel", "MLPGraphIndependent", "MLP",
           "DEFAULT_GRAPH_NET_PARAMS", "DEFAULT_ARCH", "default_params", "build_graph", "postprocess", "FeatureStore", "cross_entropy", "cross_entropy_steps", "ops"]
