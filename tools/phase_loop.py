#!/usr/bin/env python3
"""Phase-by-phase forward loop on one workload (for rocprofv3 traces of the path a multi-GPU rank runs: one operand-split pass
and ONE layer-0 GEMM launch, where the one-call forward of tools/fwd_loop.py cuts layer 0 into row panels beside the split):
    python tools/phase_loop.py cfg5 [iters]
bench_dist.predict_scaling prices a rank's kernels from the summary of this loop (profiles/rNN_cfg5_phase_kernel_stats.csv)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402
from mtmc_mpn import engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
data = bench.make_workload(name, dev)
eng = engine.ForwardEngine(model)
with torch.no_grad():
    prep = eng.prepare(data.x, data.edge_index, data.edge_attr)
    for _ in range(2):
        eng.run_phases(prep)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        eng.run_phases(prep)
    b.record()
    torch.cuda.synchronize()
print(f"{name}: {a.elapsed_time(b) / iters:.3f} ms per phase-path forward ({iters + 2} forwards profiled)")
