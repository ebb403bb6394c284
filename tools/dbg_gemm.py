#!/usr/bin/env python3
"""One encoder-style GEMM through the library vs fp64:  python tools/dbg_gemm.py M K N"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9000, 2048, 1024)
lib = _lib.load()
lib.mtmc_linear_raw.restype = C.c_int32
lib.mtmc_linear_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p]
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).cuda()
W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
b = torch.randn(N, generator=g).cuda() * 0.01
Y = torch.empty(M, N, device="cuda")
scr = torch.zeros(48, dtype=torch.int32, device="cuda")
ref = (A.double() @ W.double().t() + b.double())
for rep in range(3):
    Y.zero_()
    rc = lib.mtmc_linear_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, scr.data_ptr(), None,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    err = (Y.double() - ref).abs()
    bad = err > 1e-4 * ref.abs().max()
    print(f"rep {rep}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.2f}), bad elements {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("  bad rows mod 128 histogram (top):", torch.bincount(rows % 128, minlength=128).topk(8))
        print("  bad cols mod 128 histogram (top):", torch.bincount(cols % 128, minlength=128).topk(8))
        print("  distinct bad cols:", cols.unique().numel(), "distinct bad rows:", rows.unique().numel())
        print("  sample:", idx[:6].tolist(), [f"{float(err[i, j]):.2e}" for i, j in idx[:6].tolist()])
