#!/usr/bin/env python3
"""Where the few-row encoder kernels spend their time (csrc/gemm_few.hip built with -DFEW_STAMP=1: s_memtime per wave in two
workgroups):   bash tools/build_variant.sh stamp -DFEW_STAMP=1
               MTMC_MPN_LIB=build_ab/stamp/pkg/csrc/libmtmc_mpn.so python tools/few_stamps.py [M]
Layer 0 (M x 2048 -> 1024), per k-step: consumers (fragment reads + MFMAs issued | barrier), loaders (next step's LDS-DMA
issued | vmcnt wait for the step after this one | barrier).  Later layers: entry -> statistics landed + affine -> A rows landed + activated -> W landed +
MFMAs done -> behind the barrier -> end.  Cycles of the shader clock (s_memtime)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 450
s = torch.cuda.current_stream().cuda_stream
fn = lib.mtmc_dbg_few_stamps
fn.argtypes = [C.c_void_p, C.c_void_p]
STEPS, L0N, WVN = 40, 4 + 4 * 40, 8   # (gemm_few.hip: kFsSteps, kFsL0, kFsWave)


def run(K, N, bn):
    g = torch.Generator(device="cuda").manual_seed(K + N)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(4 * M * K + 4 * N * K + 4 * (M + N) + 1024, dtype=torch.uint8, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    if bn:
        st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
        gamma, beta = torch.rand(K, device="cuda") + 0.5, 0.3 * torch.randn(K, device="cuda")
        args = (st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr())
    else:
        args = (None, None, None)
    for _ in range(5):
        rc = lib.mtmc_linear_few_raw(A.data_ptr(), K, *args, float(M), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N,
                                     work.data_ptr(), work.numel(), st.data_ptr(), s)
        assert rc == 0
    torch.cuda.synchronize()
    l0 = (C.c_ulonglong * (2 * 8 * L0N))()
    wv = (C.c_ulonglong * (2 * 8 * WVN))()
    assert fn(l0, wv) == 0
    return torch.tensor(list(l0), dtype=torch.int64).view(2, 8, L0N), torch.tensor(list(wv), dtype=torch.int64).view(2, 8, WVN)


l0, _ = run(2048, 1024, False)
nk = 2048 // 64
for blk in range(2):
    print(f"layer 0, M={M}: workgroup slot {blk}  (waves 0-3 consumers, 4-7 loaders; cycles)")
    for w in range(8):
        t = l0[blk, w]
        e, pi, le, end = t[0].item(), t[1].item(), t[2].item(), t[3].item()
        st = t[4:4 + 4 * nk].view(nk, 4)
        per = (st[nk - 1, 3].item() - st[4, 0].item()) / (nk - 4)
        f = lambda a, b_: (st[4:nk, b_] - st[4:nk, a]).float().mean().item()
        if w < 4:
            print(f"  consumer {w}: entry->loads requested {pi - e:6d}  first barrier passed {st[0, 0].item() - e:6d}  k-step {per:7.1f} "
                  f"(reads + MFMA issue {f(0, 1):6.1f}  barrier {f(2, 3):6.1f})  loop end {le - e:6d}  epilogue {end - le:6d}  total {end - e:6d}")
        else:
            print(f"  loader   {w}: entry->prologue issued {pi - e:6d}  first barrier passed {st[0, 0].item() - e:6d}  k-step {per:7.1f} "
                  f"(issue {f(0, 1):6.1f}  vmcnt wait {f(1, 2):6.1f}  barrier {f(2, 3):6.1f})  loop end {le - e:6d}  total {end - e:6d}")
for K, N in ((1024, 512), (512, 128), (128, 32)):
    _, wv = run(K, N, True)
    nkb = 4 if (K % 128 == 0 and 2 <= K // 128 <= 8) else 1
    nw = K // (32 * nkb)
    for blk in range(2):
        print(f"layer {K}->{N}, M={M}: workgroup slot {blk}, {nw} waves")
        for w in range(min(nw, 8)):
            t = wv[blk, w]
            d = [(t[i + 1] - t[i]).item() for i in range(5)]
            print(f"  wave {w}: stats+affine {d[0]:6d}  A landed+activated {d[1]:6d}  W landed+MFMA {d[2]:6d}  barrier {d[3]:6d}  "
                  f"sum+store+stats {d[4]:6d}  total {(t[5] - t[0]).item():6d}")
