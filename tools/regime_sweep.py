#!/usr/bin/env python3
"""Forward time of 4-camera graphs of growing size (E = 12 n^2), eager and HIP-graph replay, GPU-side (events):
    python tools/regime_sweep.py 250 300 350 ...        (MTMC_MPN_LIB selects the build: few-edge / many-edge regime threshold)"""
import copy
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

dev = torch.device("cuda:0")
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()


def timed(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for n in [int(a) for a in sys.argv[1:]]:
    d = graphs.camera_graph((n, n, n, n), 2048, seed=1)
    ei = d.edge_index.to(dev)
    data = types.SimpleNamespace(x=d.x.to(dev), edge_index=ei, edge_attr=d.edge_attr.to(dev))
    with torch.no_grad():
        eager = timed(lambda: model(data), 50)
        replay = model.capture(data)
        graph = timed(replay, 50)
    print(f"n={n} N={4 * n} E={ei.shape[1]}: eager {eager * 1e3:.1f} us  graph {graph * 1e3:.1f} us", flush=True)
