#!/usr/bin/env python3
"""One encoder-style GEMM through the library vs fp64:  python tools/dbg_gemm.py M K N"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9000, 2048, 1024)
if os.environ.get("MTMC_DBG_LIB"):            # an experiment build of the library (tools/hazard_ab.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["MTMC_DBG_LIB"])
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
        # What did the block multiply instead of W[n, :]?  Within one 128-row tile the error of column n is A_tile . dW;
        # fit dW over each 64-wide k-tile (128 equations, 64 unknowns) and report the k-tile that explains it.
        if rep == 0 and os.environ.get("MTMC_DBG_FIT", "1") == "1":
            E = (Y.double() - ref)
            seen = 0
            for i, j in idx.tolist():
                tm = i // 128
                if i % 128 != 0 and seen:      # one (tile, col) per sample
                    continue
                rows_t = slice(tm * 128, min(tm * 128 + 128, M))
                e = E[rows_t, j]
                At = A[rows_t].double()
                best = None
                for kt in range(K // 64):
                    Ak = At[:, kt * 64:(kt + 1) * 64]
                    d = torch.linalg.lstsq(Ak, e.unsqueeze(1)).solution[:, 0]
                    res = float((Ak @ d - e).norm() / e.norm())
                    if best is None or res < best[0]:
                        best = (res, kt, d)
                res, kt, d = best
                w = W[j, kt * 64:(kt + 1) * 64].double()
                print(f"  tile {tm} col {j}: |e| {float(e.norm()):.3e}; best k-tile {kt} leaves {res:.2e} of it; "
                      f"dW.W/W.W = {float(d @ w / (w @ w)):+.4f}, |dW|/|W| = {float(d.norm() / w.norm()):.4f}")
                ratio = (d / w).reshape(16, 4)      # one row per storing lane (4 consecutive k each)
                print("    dW/W per lane (rows) x k (cols):", " | ".join(" ".join(f"{float(v):+.2f}" for v in r) for r in ratio))
                print("    column means of dW/W:", [f"{float(v):+.6f}" for v in ratio.mean(0)], " (1/256 - 1 = -0.996094)")
                seen += 1
                if seen >= 6:
                    break
