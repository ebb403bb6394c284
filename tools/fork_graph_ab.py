#!/usr/bin/env python3
"""S02 forward: eager / HIP-graph replay, each with and without MTMC_F_FORK (edge branch on a side stream: graph edges under
capture, events in eager mode):  python tools/fork_graph_ab.py"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402
from mtmc_mpn import _lib, modules  # noqa: E402

dev = torch.device("cuda:0")
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
data = bench.make_workload("s02", dev)
for fork in (0, 1, 0, 1):
    torch.manual_seed(0)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
    if fork:                                   # the module's only route for a library flag is `deterministic`: borrow it
        modules._lib_flags.F_DETERMINISTIC = _lib.F_FORK
        model.deterministic = True
    else:
        modules._lib_flags.F_DETERMINISTIC = 1
    sec, d = bench.time_forward(model, data, 200, 30)
    with torch.no_grad():
        replay = model.capture(data)
    rsec, rd = bench.time_forward(model, data, 200, 30, fn=replay)
    print(f"fork={fork}: eager {sec * 1e6:.1f} us (median {d['median'] * 1e3:.1f})   graph replay {rsec * 1e6:.1f} us (median {rd['median'] * 1e3:.1f})")
