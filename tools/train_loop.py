#!/usr/bin/env python3
"""Training-step loop (config 3 shape) for rocprofv3 traces:  python tools/train_loop.py [iters]"""
import copy
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "tests", "golden", "train_tracklets.json")) as f:
    tr = json.load(f)["tracklets"]
d = graphs.training_graph(tr, 100, 2048, 3)
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=3)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)   # one kernel for all 34 tensors
ei = d.edge_index.t().contiguous().to(dev).t()
data = types.SimpleNamespace(x=d.x.to(dev), edge_index=ei, edge_attr=d.edge_attr.to(dev))
labels = d.edge_labels.long().to(dev)


def step(parts=None):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    out, _ = model(data)
    if parts is not None:
        torch.cuda.synchronize(); t.append(time.perf_counter())
    loss = mtmc_mpn.cross_entropy_steps(out["classified_edges"], labels)
    if parts is not None:
        torch.cuda.synchronize(); t.append(time.perf_counter())
    loss.backward()
    if parts is not None:
        torch.cuda.synchronize(); t.append(time.perf_counter())
    opt.step()
    if parts is not None:
        torch.cuda.synchronize(); t.append(time.perf_counter())
        for i in range(4):
            parts[i] += t[i + 1] - t[i]


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print(f"train step: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms  (N={data.x.shape[0]}, E={ei.shape[1]})")
parts = [0.0] * 4
for _ in range(iters):
    step(parts)
print("synchronised split, ms: forward %.3f  loss %.3f  backward %.3f  optimizer %.3f" % tuple(p / iters * 1e3 for p in parts))
print(f"steps profiled: {5 + 2 * iters}")
