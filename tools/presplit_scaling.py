#!/usr/bin/env python3
"""gemm_f16p_m16_kernel (layer 0 of many-row graphs: planes x planes) alone, for growing M: time per 100k rows.
    python tools/presplit_scaling.py [M ...]     (reuse_planes = 1: the operand split is not part of the timed launch)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

lib = _lib.load()
K, N = 2048, 1024
s = torch.cuda.current_stream().cuda_stream
for M in [int(a) for a in sys.argv[1:]] or [100_000, 200_000, 400_000, 1_000_000]:
    A = torch.randn(M, K, device="cuda")
    W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(M * K * 4 + N * K * 4 + (M + N) * 4 + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")

    def run(reuse):
        rc = lib.mtmc_linear_presplit_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(),
                                          work.numel(), scr.data_ptr(), st.data_ptr(), reuse, s)
        assert rc == 0, rc
    run(0)
    for _ in range(3):
        run(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10):
            run(1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"M={M}: {best:.3f} ms = {best / M * 1e5:.3f} ms per 100k rows = {2.0 * M * K * N / best / 1e9:.0f} TFLOP/s fp32-equivalent "
          f"({3 * 2.0 * M * K * N / best / 1e9 / 2500:.2f} of 2.5 PF in fp16 products)", flush=True)
    del A, W, Y, work
