#!/usr/bin/env python3
"""The config-3 training step as one HIP graph, replayed (for rocprofv3 --kernel-trace):  python tools/train_graph_loop.py [replays]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
model, opt, data, labels, ce_weight = bench.training_setup(dev)
replay = model.capture_training_step(
    data, lambda o, _h: mtmc_mpn.cross_entropy_steps(o["classified_edges"], labels, weight=ce_weight), opt)
for _ in range(10):
    replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    replay()
torch.cuda.synchronize()
print(f"graph-replayed train step: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms")
print(f"steps profiled: {3 + 10 + reps}")
