#!/usr/bin/env python3
"""Timeline of one workgroup of the role-split GEMM (csrc/gemm_staged.hip built with -DSG_STAMP=1: s_memtime at four points
of every k-tile, per wave):   make -C <csrc> EXTRA=-DSG_STAMP=1 libmtmc_mpn.so && python tools/staged_stamps.py [M K N]
Points: 0 k-tile start (behind the barrier), 1 work issued, 2 own waits done (producer: ds_write landed; consumer: fragment
reads + its W DMA share of the next k-tile landed), 3 behind the barrier.  Prints, per wave, the mean cycles spent in work
(0->1), in its own waits (1->2) and at the barrier (2->3) over the steady-state k-tiles."""
import ctypes as C
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
for _ in range(3):
    rc = lib.mtmc_linear_staged_raw(A.data_ptr(), K, st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(M), W.data_ptr(),
                                    b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(), scr.data_ptr(), None, s)
    assert rc == 0
torch.cuda.synchronize()
KT = 64
buf = (C.c_ulonglong * (12 * KT * 4))()
fn = lib.mtmc_dbg_staged_stamps
fn.argtypes = [C.c_void_p]
assert fn(buf) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(12, KT, 4)
nk = min(K // 32, KT)
print(f"M={M} K={K} N={N}: {nk} k-tiles recorded; mean cycles per k-tile and of its parts, steady state")
for w in range(12):
    role = "producer" if w < 4 else "consumer"
    tt = t[w, 2:nk - 2]
    per = (t[w, nk - 3, 0] - t[w, 2, 0]).item() / (nk - 5)
    f = lambda a, b: (tt[:, b] - tt[:, a]).float().mean().item()
    print(f"wave {w:2d} {role}: k-tile {per:7.1f}  work {f(0, 1):7.1f}  own waits {f(1, 2):6.1f}  barrier {f(2, 3):7.1f}")
