#!/usr/bin/env python3
"""Where does the host time of one training step go?  (config 3: forward with tape + loss + backward + SGD)
    python tools/train_host_profile.py [--profile]"""
import copy
import cProfile
import json
import os
import pstats
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

dev = torch.device("cuda:0")
with open(os.path.join(ROOT, "tests", "golden", "train_tracklets.json")) as f:
    tr = json.load(f)["tracklets"]
d = graphs.training_graph(tr, 100, 2048, 3)
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=3)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
ei = d.edge_index.t().contiguous().to(dev).t()
data = types.SimpleNamespace(x=d.x.to(dev), edge_index=ei, edge_attr=d.edge_attr.to(dev))
labels = d.edge_labels.long().to(dev)
n1 = float(labels.sum())
w = torch.tensor([1.0, (labels.numel() - n1) / max(n1, 1.0)], device=dev)
T = {"zero": 0.0, "fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0}


def step(timed=False):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    out, _ = model(data)
    t2 = time.perf_counter()
    loss = mtmc_mpn.cross_entropy_steps(out["classified_edges"], labels, weight=w)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if timed:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[k] += v
    return loss


for _ in range(10):
    step()
torch.cuda.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    step(True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"wall {t_all / reps * 1e3:.3f} ms/step; host issue {t_issue / reps * 1e3:.3f} ms/step; host by part (us):",
      {k: round(v / reps * 1e6, 1) for k, v in T.items()})
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    step()
e1.record()
torch.cuda.synchronize()
print(f"GPU-side elapsed {e0.elapsed_time(e1) / reps:.3f} ms/step")
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
