#!/usr/bin/env python3
"""Plain forward loop on one workload (for rocprofv3 traces):  python tools/fwd_loop.py s02 [iters]
(MTMC_DETERMINISTIC=1: the module's fixed-order aggregation mode)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s02"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
model.deterministic = bool(os.environ.get("MTMC_DETERMINISTIC"))
data = bench.make_workload(name, dev)
sec, _ = bench.time_forward(model, data, iters, 10)
print(f"{name}: {sec * 1e6:.1f} us/forward")
