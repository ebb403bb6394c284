#!/usr/bin/env python3
"""Where the round kernels of a few-edge forward spend their time (library built with -DEK_STAMP=1: s_memtime at fixed points of
workgroup 0 and of the middle workgroup, thread 0):
    bash tools/build_variant.sh stamp "-DFEW_STAMP=1 -DEK_STAMP=1"
    MTMC_MPN_LIB=build_ab/stamp/pkg/csrc/libmtmc_mpn.so python tools/edge_stamps.py [workload]
Prints, per kernel (last round of the forward; pass A also the first-round form), the cycles between consecutive points."""
import ctypes as C
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402
from mtmc_mpn import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s02"
lib = _lib.load()
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
data = bench.make_workload(name, dev)
with torch.no_grad():
    for _ in range(20):
        model(data)
torch.cuda.synchronize()
SEG = {
    "node_proj_kernel": (0, "node", ["weights staged + barrier", "h rows staged + barrier", "FMAs + P/Q stores", "z2 sums -> atomics", "(block 0: edge-encoder affines)"]),
    "node_proj_mfma_kernel": (5, "node", ["weights + y statistics + barrier", "rows -> MFMA -> P/Q stores", "z2 sums -> atomics", "(block 0: edge-encoder affines)"]),
    "pass_a_kernel (later round)": (1, "edge", ["consts staged + barrier", "loads -> z1 stored", "block sums -> atomics"]),
    "pass_a_kernel (first round)": (2, "edge", ["consts staged + barrier", "loads -> z1 stored", "block sums -> atomics"]),
    "pass_b_kernel": (3, "edge", ["loads issued, z1 stats gathered + barrier", "affine + barrier", "e' stored, moments, run sums", "block sums -> atomics"]),
    "pass_c_mfma_kernel": (4, "edge", ["stats gathered + barrier", "affine of 32 channels", "chunks: MFMA, sums, atomics"]),
}
bufs = {}
for tu in ("edge", "node"):
    fn = getattr(lib, f"mtmc_dbg_ek_stamps_{tu}")
    fn.argtypes = [C.c_void_p]
    b = (C.c_ulonglong * (8 * 2 * 8))()
    assert fn(b) == 0
    bufs[tu] = torch.tensor(list(b), dtype=torch.int64).view(8, 2, 8)
for kname, (kid, tu, segs) in SEG.items():
    for slot, what in ((0, "workgroup 0"), (1, "middle workgroup")):
        t = bufs[tu][kid, slot]
        d = [(t[i + 1] - t[i]).item() for i in range(len(segs))]
        print(f"{kname:30s} {what:16s} total {(t[len(segs)] - t[0]).item():6d} cyc: " + " | ".join(f"{s} {v}" for s, v in zip(segs, d)))
