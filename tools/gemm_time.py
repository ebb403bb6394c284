#!/usr/bin/env python3
"""Time one encoder-style GEMM through the library:  python tools/gemm_time.py M K N [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100000, 2048, 1024)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _lib.load()
A = torch.randn(M, K, device="cuda")
W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
b = torch.zeros(N, device="cuda")
Y = torch.empty(M, N, device="cuda")
scr = torch.zeros(48, dtype=torch.int32, device="cuda")
st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.mtmc_linear_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, scr.data_ptr(), st.data_ptr(), s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    lib.mtmc_linear_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, scr.data_ptr(), st.data_ptr(), s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"M={M} K={K} N={N}: {ms * 1e3:.1f} us per call (incl. the |.|max pass and memsets), {2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
