#!/usr/bin/env python3
"""Is the role-split GEMM (layer 1 of config 4) limited by the board's power cap?  The same instruction stream on random and on
all-zero operands (zero operands toggle almost nothing in the matrix cores and the LDS): if the zero run is markedly faster,
the kernel is energy-bound -- what runs beside the MFMAs is paid for in clock, not hidden.  Also the layer-0 kernel for
comparison (DESIGN.md A.2 measured it with clock / power sampling).   python tools/staged_zero_vs_random.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream


def time_it(fn, reps=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def staged(M, K, N, zero):
    A = torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda") * 2 + 0.5
    gamma = torch.zeros(K, device="cuda") if zero else torch.rand(K, device="cuda") + 0.5
    beta = torch.zeros(K, device="cuda") if zero else 0.3 * torch.randn(K, device="cuda")
    W = torch.zeros(N, K, device="cuda") if zero else (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    b = torch.zeros(N, device="cuda")
    st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(4 * N * K + 4 * N + 512, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")

    def run():
        rc = lib.mtmc_linear_staged_raw(A.data_ptr(), K, st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(M), W.data_ptr(),
                                        b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(), scr.data_ptr(), st.data_ptr(), s)
        assert rc == 0
    return time_it(run)


def presplit(M, K, N, zero):
    A = torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda")
    W = torch.zeros(N, K, device="cuda") if zero else (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(4 * M * K + 4 * N * K + 4 * (M + N) + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    args = (A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(), scr.data_ptr(), st.data_ptr())
    lib.mtmc_linear_presplit_raw(*args, 0, s)

    def run():                                   # the GEMM alone on the planes already in `work`
        assert lib.mtmc_linear_presplit_raw(*args, 1, s) == 0
    return time_it(run)


for name, fn, shape in (("role-split GEMM, layer 1 (whole raw call: |A|max + W split + GEMM)", staged, (100000, 1024, 512)),
                        ("pre-split GEMM, layer 0 (GEMM alone)", presplit, (100000, 2048, 1024))):
    r, z = fn(*shape, False), fn(*shape, True)
    r2, z2 = fn(*shape, False), fn(*shape, True)
    print(f"{name} {shape}: random {r:.3f} / {r2:.3f} ms, zeros {z:.3f} / {z2:.3f} ms  -> zeros / random = {(z + z2) / (r + r2):.2f}")
