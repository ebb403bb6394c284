"""Loader of the kernel laboratory (csrc/lab/ -> libmtmc_lab.so): A/B variants and timing experiments of the pre-split GEMM
(some with deliberately wrong results) and the second form of the role-split GEMM (lab/staged2_lab.hip).  Test / tool infrastructure: used by tests/test_gpu_gemm_presplit.py,
tools/presplit_time.py and tools/gemm_clock_watch.py -- the product package has no reference to it."""
import ctypes as C
import os

from mtmc_mpn import _lib

LAB_PATH = os.environ.get("MTMC_LAB_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libmtmc_lab.so")   # (variant builds: A/B)
_lab = None


def load_lab() -> C.CDLL:
    global _lab
    if _lab is None:
        _lib.load()                                  # the laboratory links the product's operand-split pass
        if not os.path.exists(LAB_PATH):
            raise RuntimeError(f"{LAB_PATH} missing; run `python -m mtmc_mpn.build`")
        _lab = C.CDLL(LAB_PATH)
        _lab.mtmc_lab_linear_presplit_raw.restype = C.c_int32
        _lab.mtmc_lab_linear_presplit_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                      C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                      C.c_int32, C.c_void_p]
        _lab.mtmc_lab_linear_staged2_raw.restype = C.c_int32
        _lab.mtmc_lab_linear_staged2_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                     C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lab
