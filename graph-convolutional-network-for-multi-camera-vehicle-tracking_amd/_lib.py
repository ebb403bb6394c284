"""ctypes binding of the C ABI in include/mtmc_mpn.h (csrc/libmtmc_mpn.so, built for gfx950).

The library is built in-tree (`python -m mtmc_mpn.build`, or `__graft_entry__.build()`).  If it is
missing, loading fails loudly: there is no other implementation behind the module.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_ENC_LAYERS = 8
NODE_DIM, EDGE_DIM, MAX_CLASSES = 32, 4, 4
AGG = {"sum": 0, "mean": 1, "max": 2}

(PH_BEGIN, PH_EDGE_ENC, PH_NODE_ENC, PH_NODE_H0, PH_ROUND_PROJ, PH_ROUND_A, PH_ROUND_B, PH_ROUND_STAT,
 PH_ROUND_C, PH_END, PH_NODE_COMBINE) = range(11)

E_ARG, E_WORKSPACE, E_HIP, E_ROWS = -1, -2, -3, -4

_f32p = C.c_void_p


class Layer(C.Structure):
    _fields_ = [("weight", _f32p), ("bias", _f32p), ("gamma", _f32p), ("beta", _f32p),
                ("in_dim", C.c_int32), ("out_dim", C.c_int32)]


class Model(C.Structure):
    """mtmc_mpn_model.  struct_bytes (since ABI v5) is filled in on construction: the library refuses any other size."""
    _fields_ = [("struct_bytes", C.c_uint32), ("n_enc_layers", C.c_int32), ("enc_node", Layer * MAX_ENC_LAYERS), ("enc_edge", Layer * 2),
                ("upd_edge", Layer), ("upd_node", Layer), ("cls", Layer),
                ("agg", C.c_int32), ("num_enc_steps", C.c_int32), ("num_class_steps", C.c_int32),
                ("reattach_nodes", C.c_int32), ("reattach_edges", C.c_int32),
                ("dropout_enc", C.c_float), ("dropout_upd_edge", C.c_float), ("dropout_upd_node", C.c_float)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_bytes = C.sizeof(type(self))


class Call(C.Structure):
    """mtmc_mpn_call.  struct_bytes is filled in on construction, as in Model."""
    _fields_ = [("struct_bytes", C.c_uint32), ("x", C.c_void_p), ("x_row_stride", C.c_int64), ("row", C.c_void_p), ("col", C.c_void_p),
                ("idx_stride", C.c_int64), ("edge_attr", C.c_void_p), ("n_nodes", C.c_int64), ("n_edges", C.c_int64),
                ("n_edges_total", C.c_int64), ("node_lo", C.c_int64), ("node_hi", C.c_int64),
                ("logits", C.c_void_p), ("h_out", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("training", C.c_int32), ("flags", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_void_p), ("row_lo", C.c_int64), ("row_hi", C.c_int64),
                ("weight_cache", C.c_void_p), ("weight_cache_bytes", C.c_size_t)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_bytes = C.sizeof(type(self))


class WsLayout(C.Structure):
    _fields_ = [("total_bytes", C.c_size_t), ("zero_bytes", C.c_size_t), ("flags_off", C.c_size_t),
                ("stat_attr_off", C.c_size_t), ("stat_enc2_off", C.c_size_t), ("stat_enc_layer_off", C.c_size_t * MAX_ENC_LAYERS),
                ("stat_round_off", C.c_size_t), ("deg_off", C.c_size_t), ("seg_off", C.c_size_t),
                ("h0_off", C.c_size_t), ("h_acc_off", C.c_size_t * 2), ("deg_global_off", C.c_size_t),
                ("P_off", C.c_size_t)]


class Plan(C.Structure):
    """mtmc_mpn_plan: which kernels a call would run (host-only query)."""
    _fields_ = [("enc_kernel", C.c_int32 * MAX_ENC_LAYERS), ("enc_split_k", C.c_int32 * MAX_ENC_LAYERS),
                ("edges_per_thread", C.c_int32), ("lazy_edges", C.c_int32), ("pass_c", C.c_int32),
                ("avg_degree", C.c_double), ("pass_a_col_blocks", C.c_int32), ("layer0_panels", C.c_int32),
                ("enc2_passenger", C.c_int32), ("node_stat_folded", C.c_int32)]


(GEMM_GENERIC, GEMM_INLOOP_64, GEMM_INLOOP_128, GEMM_PRESPLIT_256, GEMM_STAGED_128, GEMM_ROWS_16, GEMM_FEW_L0,
 GEMM_FEW_WAVE) = range(8)
PASS_C_WALK, PASS_C_MFMA_SORTED, PASS_C_MFMA_ANY = range(3)

# statistics blocks (include/mtmc_mpn.h): replicas x stride doubles each
STAT_REPLICAS, ATTR_STRIDE, ENC2_STRIDE, Z1_STRIDE, M_STRIDE, Z2_STRIDE = 16, 16, 16, 16, 16, 64
ROUND_BLOCK = STAT_REPLICAS * (Z1_STRIDE + M_STRIDE + Z2_STRIDE)
F_DETERMINISTIC, F_GLOBAL_DEG, F_FORK, F_SEED_ON_DEVICE = 1, 4, 2, 16

# MTMC_MPN_LIB: another build of the same ABI (same-box A/B of two library versions, tools/lib_ab.sh); default: the in-tree build
LIB_PATH = os.environ.get("MTMC_MPN_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmtmc_mpn.so")
EXPORTS = ["mtmc_mpn_abi_version", "mtmc_mpn_last_error", "mtmc_mpn_workspace_bytes", "mtmc_mpn_workspace_layout",
           "mtmc_mpn_forward", "mtmc_mpn_run_phase", "mtmc_mpn_run_phases", "mtmc_mpn_plan_call", "mtmc_scatter_add", "mtmc_scatter_add_i64", "mtmc_scatter_mean", "mtmc_scatter_max",
           "mtmc_mlp_layer_forward", "mtmc_mpn_train_workspace_bytes", "mtmc_mpn_backward", "mtmc_graph_workspace_bytes",
           "mtmc_build_graph", "mtmc_postprocess_workspace_bytes", "mtmc_postprocess",
           "mtmc_cross_entropy_forward", "mtmc_cross_entropy_backward",
           "mtmc_cross_entropy_steps_forward", "mtmc_cross_entropy_steps_backward", "mtmc_mpn_backward_steps", "mtmc_mpn_backward_flat", "mtmc_mpn_grad_layout", "mtmc_linear_raw", "mtmc_edge_confusion",
           "mtmc_linear_presplit_raw", "mtmc_linear_staged_raw", "mtmc_linear_few_raw", "mtmc_mpn_weight_cache_bytes"]

_lib = None


def load() -> C.CDLL:
    """dlopen the HIP library (after torch, so both share torch's libamdhip64) and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"mtmc_mpn: HIP extension not built ({LIB_PATH} missing); run `python -m mtmc_mpn.build` "
                           "-- there is no fallback implementation")
    import torch  # noqa: F401  (loads the ROCm runtime the library links against)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.mtmc_mpn_abi_version.restype = C.c_int32
    lib.mtmc_mpn_last_error.restype = C.c_char_p
    lib.mtmc_mpn_workspace_bytes.restype = C.c_size_t
    lib.mtmc_mpn_workspace_bytes.argtypes = [C.POINTER(Model), C.c_int64, C.c_int64]
    lib.mtmc_mpn_weight_cache_bytes.restype = C.c_size_t
    lib.mtmc_mpn_weight_cache_bytes.argtypes = [C.POINTER(Model)]
    lib.mtmc_linear_few_raw.restype = C.c_int32
    lib.mtmc_linear_few_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p]
    lib.mtmc_mpn_workspace_layout.restype = C.c_int32
    lib.mtmc_mpn_workspace_layout.argtypes = [C.POINTER(Model), C.c_int64, C.c_int64, C.POINTER(WsLayout)]
    lib.mtmc_mpn_forward.restype = C.c_int32
    lib.mtmc_mpn_forward.argtypes = [C.POINTER(Model), C.POINTER(Call)]
    lib.mtmc_mpn_train_workspace_bytes.restype = C.c_size_t
    lib.mtmc_mpn_train_workspace_bytes.argtypes = [C.POINTER(Model), C.c_int64, C.c_int64]
    lib.mtmc_mpn_backward.restype = C.c_int32
    lib.mtmc_mpn_backward.argtypes = [C.POINTER(Model), C.POINTER(Call), C.c_void_p, C.c_void_p, C.POINTER(Model),
                                      C.c_void_p, C.c_void_p]
    lib.mtmc_graph_workspace_bytes.restype = C.c_size_t
    lib.mtmc_graph_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
    lib.mtmc_build_graph.restype = C.c_int32
    lib.mtmc_build_graph.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + \
        [C.c_int32, C.c_int64] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]
    lib.mtmc_mpn_backward_steps.restype = C.c_int32
    lib.mtmc_mpn_backward_steps.argtypes = [C.POINTER(Model), C.POINTER(Call), C.POINTER(C.c_void_p), C.c_void_p,
                                            C.POINTER(Model), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.mtmc_mpn_backward_flat.restype = C.c_int32
    lib.mtmc_mpn_backward_flat.argtypes = [C.POINTER(Model), C.POINTER(Call), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p]
    lib.mtmc_mpn_grad_layout.restype = C.c_int64
    lib.mtmc_mpn_grad_layout.argtypes = [C.POINTER(Model), C.POINTER(C.c_int64), C.c_int32]
    lib.mtmc_edge_confusion.restype = C.c_int32
    lib.mtmc_edge_confusion.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mtmc_linear_presplit_raw.restype = C.c_int32
    lib.mtmc_linear_presplit_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.mtmc_linear_staged_raw.restype = C.c_int32
    lib.mtmc_linear_staged_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_linear_raw.restype = C.c_int32
    lib.mtmc_linear_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_cross_entropy_forward.restype = C.c_int32
    lib.mtmc_cross_entropy_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_cross_entropy_backward.restype = C.c_int32
    lib.mtmc_cross_entropy_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_cross_entropy_steps_forward.restype = C.c_int32
    lib.mtmc_cross_entropy_steps_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                     C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_cross_entropy_steps_backward.restype = C.c_int32
    lib.mtmc_cross_entropy_steps_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                      C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtmc_postprocess_workspace_bytes.restype = C.c_size_t
    lib.mtmc_postprocess_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    lib.mtmc_postprocess.restype = C.c_int32
    lib.mtmc_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]
    lib.mtmc_mpn_plan_call.restype = C.c_int32
    lib.mtmc_mpn_plan_call.argtypes = [C.POINTER(Model), C.POINTER(Call), C.POINTER(Plan)]
    lib.mtmc_mpn_run_phase.restype = C.c_int32
    lib.mtmc_mpn_run_phase.argtypes = [C.POINTER(Model), C.POINTER(Call), C.c_int32, C.c_int32]
    lib.mtmc_mpn_run_phases.restype = C.c_int32
    lib.mtmc_mpn_run_phases.argtypes = [C.POINTER(Model), C.POINTER(Call), C.POINTER(C.c_int32), C.c_int32]
    for name in ("mtmc_scatter_add", "mtmc_scatter_add_i64"):
        getattr(lib, name).restype = C.c_int32
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.mtmc_scatter_mean.restype = C.c_int32
    lib.mtmc_scatter_mean.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    lib.mtmc_scatter_max.restype = C.c_int32
    lib.mtmc_scatter_max.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    lib.mtmc_mlp_layer_forward.restype = C.c_int32
    lib.mtmc_mlp_layer_forward.argtypes = [C.POINTER(Layer), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    if lib.mtmc_mpn_abi_version() != 6:
        raise RuntimeError("mtmc_mpn: ABI version mismatch between _lib.py and libmtmc_mpn.so")
    _lib = lib
    return lib


def check(rc: int):
    """Map a negative return code to the exception the reference's path would raise."""
    if rc == 0:
        return
    msg = load().mtmc_mpn_last_error().decode()
    if rc == E_ROWS:
        raise ValueError(msg)
    raise RuntimeError(f"mtmc_mpn: {msg} (code {rc})")
