#!/usr/bin/env python3
"""What a library fp16 GEMM of layer 0's shape reaches on this box (a yardstick for gemm_f16p_m16_kernel, not a product path):
the three-product form is ONE fp16 GEMM with K' = 3K -- [a1 a1 a2] . [w1 w2 w1]^T -- here M x 1024 x 6144 through torch.mm (hipBLASLt /
rocBLAS), random data (the power cap is data dependent), fp16 output.   python tools/gemm_yardstick.py [M ...]"""
import sys

import torch

dev = torch.device("cuda:0")
for m in [int(a) for a in sys.argv[1:]] or [100_000]:
    for n, k in ((1024, 2048), (1024, 6144), (512, 3072)):      # layer 0 (K, 3K); layer 1 (3K)
        a = torch.randn(m, k, device=dev, dtype=torch.float16)
        w = torch.randn(n, k, device=dev, dtype=torch.float16)
        wt = w.t()
        for _ in range(5):
            torch.mm(a, wt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(20):
                torch.mm(a, wt)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        tf = 2.0 * m * n * k / (best * 1e-3) / 1e12
        print(f"M={m} N={n} K={k}: {best:.3f} ms = {tf:.0f} TFLOP/s fp16 ({tf / 2500:.2f} of 2.5 PF)", flush=True)
        del a, w, wt
