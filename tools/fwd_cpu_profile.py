#!/usr/bin/env python3
"""Host-side profile of the eval forward (cProfile):  python tools/fwd_cpu_profile.py [workload]"""
import copy
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s02"
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)), None, "resnet101").to(dev).eval()
data = bench.make_workload(name, dev)
with torch.no_grad():
    for _ in range(20):
        model(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        model(data)
    t1 = time.perf_counter()                      # host time to ISSUE 300 forwards (no sync yet)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e6 * (t1 - t0) / 300:.1f} us per forward on the host; complete {1e6 * (t2 - t0) / 300:.1f} us")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        model(data)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
