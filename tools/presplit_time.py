#!/usr/bin/env python3
"""Pre-split first-layer GEMM (csrc/gemm_presplit.hip) against float64 and against the in-loop-split kernel:
   python tools/presplit_time.py [M K N]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import lab_lib  # noqa: E402  (the kernel laboratory's loader lives with the tests)

lib = _lib.load()
lab = lab_lib.load_lab()            # the A/B variants live in the kernel laboratory (csrc/lab/, libmtmc_lab.so)
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,9").split(",")]
s = torch.cuda.current_stream().cuda_stream


def run(M, K, N, variant, A, W, b, Y, work, scr, st):
    rc = lab.mtmc_lab_linear_presplit_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(),
                                          work.numel(), scr.data_ptr(), st.data_ptr(), variant, s)
    assert rc == 0, rc


def check(M, K, N):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, K, device="cuda", generator=g) * torch.exp(3 * torch.randn(M, 1, device="cuda", generator=g))
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    ref = (A.double() @ W.double().t() + b.double())
    work = torch.empty(M * K * 4 + N * K * 4 + (M + N) * 4 + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    scale = (A.double().abs() @ W.double().abs().t()) + b.double().abs()          # per-element sum |a||w|
    for variant in (0, 2, 3, 4, 8, 9, 10, 11):  # 12-17 are timing experiments with wrong results
        Y = torch.full((M, N), float("nan"), device="cuda")
        run(M, K, N, variant, A, W, b, Y, work, scr, st)
        torch.cuda.synchronize()
        err = ((Y.double() - ref).abs() / scale).max().item()
        serr = (st[:N] - ref.sum(0)).abs().max().item() / ref.abs().sum(0).max().item()
        ymax = scr[32:48].view(torch.float32).max().item()
        print(f"  M={M} K={K} N={N} variant {variant}: max |err| / sum|a||w| = {err:.2e}  stats rel {serr:.1e}  "
              f"ymax {ymax:.4g} vs {Y.abs().max().item():.4g}", flush=True)
        if not (err < 3e-7 and torch.isfinite(Y).all()):
            E = ((Y.double() - ref).abs() / scale)
            bad = E > 3e-7
            print("    bad fraction", bad.float().mean().item(), "rows with bad", bad.any(1).sum().item(), "cols with bad",
                  bad.any(0).sum().item())
            rows = bad.any(1).nonzero().flatten()[:24].tolist()
            cols = bad.any(0).nonzero().flatten()[:24].tolist()
            print("    first bad rows", rows, "cols", cols)
            print("    row max err by row%32:", [f"{E[r::32].max().item():.1e}" for r in range(0, 32, 4)])


def timeit(M, K, N, iters=20):
    A = torch.randn(M, K, device="cuda")
    W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    if os.environ.get("DATA") == "zeros":        # same instruction stream on all-zero operands: what the DATA costs (power)
        A.zero_(), W.zero_()
    elif os.environ.get("DATA") == "ones":
        A.fill_(1.0), W.fill_(1.0 / 64)
    b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(M * K * 4 + N * K * 4 + (M + N) * 4 + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    base = t(lambda: lib.mtmc_linear_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, scr.data_ptr(),
                                         st.data_ptr(), s))
    print(f"M={M} K={K} N={N}: in-loop split (|.|max pass + GEMM) {base:.1f} us", flush=True)
    for variant in VARIANTS:
        full = t(lambda: run(M, K, N, variant, A, W, b, Y, work, scr, st))
        gemm = t(lambda: run(M, K, N, -variant - 1, A, W, b, Y, work, scr, st))
        print(f"  variant {variant}: split + GEMM {full:.1f} us, GEMM alone {gemm:.1f} us = "
              f"{2.0 * M * K * N / gemm / 1e6:.1f} TFLOP/s fp32-equivalent", flush=True)
        if variant in (17, 19, 20):                                         # per-wave phase averages (shader clocks per k-tile)
            torch.cuda.synchronize()
            ph = Y[0, :32].reshape(8, 4).cpu()
            for w in range(8):
                print(f"    wave {w}: own DMA wait {ph[w, 0]:.0f}  barrier {ph[w, 1]:.0f}  DMA issue {ph[w, 2]:.0f}  "
                      f"reads+MFMA issue {ph[w, 3]:.0f}", flush=True)


if len(sys.argv) > 3:
    M, K, N = (int(a) for a in sys.argv[1:4])
    check(min(M, 3000), K, N)
    timeit(M, K, N)
else:
    for shape in [(129, 64, 128), (300, 512, 130), (1000, 2048, 1024)]:
        check(*shape)
    timeit(100000, 2048, 1024)
    timeit(20000, 2048, 1024)
