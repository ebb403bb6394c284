#!/usr/bin/env python3
"""Forward captured in a HIP graph vs eager launches:  python tools/graph_replay.py [workload] [iters]"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s02"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
data = bench.make_workload(name, dev)
eager = bench.time_forward(model, data, iters, 20)
with torch.no_grad():
    ref, ref_h = model(data)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(data)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, h = model(data)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / iters
    ok = all(torch.allclose(a, b, atol=1e-5) for a, b in zip(out["classified_edges"], ref["classified_edges"]))
print(f"{name}: eager {eager * 1e6:.1f} us, graph replay {rep * 1e6:.1f} us, same logits: {ok}")
