#!/usr/bin/env python3
"""Encoder layers >= 1 in isolation: the role-split kernel (gemm_staged.hip) against float64 and in time.
   python tools/staged_time.py [M K N]   (default: layer 1 of config 4, 100000 1024 512)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

lib = _lib.load()
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (100000, 1024, 512)
s = torch.cuda.current_stream().cuda_stream
A = torch.randn(M, K, device="cuda") * 2 + 0.5
gamma, beta = torch.rand(K, device="cuda") + 0.5, 0.3 * torch.randn(K, device="cuda")
W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
b = torch.randn(N, device="cuda")
st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
Y = torch.empty(M, N, device="cuda")
work = torch.empty(4 * N * K + 4 * N + 512, dtype=torch.uint8, device="cuda")
scr = torch.zeros(48, dtype=torch.int32, device="cuda")
st = torch.empty(2 * N, dtype=torch.float64, device="cuda")


fn = lib.mtmc_linear_staged_raw
if os.environ.get("STAGED_LAB", "0") not in ("", "0"):       # the kernel laboratory's second form (csrc/lab/staged2_lab.hip)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import lab_lib  # noqa: E402
    fn = lab_lib.load_lab().mtmc_lab_linear_staged2_raw


def run():
    rc = fn(A.data_ptr(), K, st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(M), W.data_ptr(),
            b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(), scr.data_ptr(), st.data_ptr(), s)
    assert rc == 0, lib.mtmc_mpn_last_error()


run()
torch.cuda.synchronize()
mean = st_in[:K] / M
var = (st_in[K:] / M - mean * mean).clamp_min(0)
rows = torch.arange(0, M, max(1, M // 4096), device="cuda")
sv = torch.rsqrt(var + 1e-5) * gamma.double()
tv = beta.double() - mean * sv
a = torch.relu(A[rows].double() * sv + tv)
ref = a @ W.double().t() + b.double()
bound = 3e-7 * (a.abs() @ W.double().abs().t() + b.double().abs()) + ((A[rows].double().abs() * sv.abs() + tv.abs()) * 2.0 ** -23) @ W.double().abs().t()
print(f"M={M} K={K} N={N}: max |err| / budget = {((Y[rows].double() - ref).abs() / bound).max().item():.2f} on {rows.numel()} rows (must be < 1)")
for _ in range(3):
    run()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
fl = 2.0 * M * K * N
print(f"  whole call (|A|max + W split + GEMM): median {ts[10]:.3f} ms, min {ts[0]:.3f} ms -> {fl / ts[10] / 1e9:.0f} TFLOP/s fp32-equivalent "
      f"= {fl / ts[10] / 1e9 / 833.3:.2f} of (fp16 peak / 3)")
