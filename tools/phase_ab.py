#!/usr/bin/env python3
"""Per-phase kernel times of one workload, for A/B runs under different env knobs:  python tools/phase_ab.py cfg4 [iters]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
_, L, cs = bench.WORKLOADS[name]
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)), None, "resnet101").to(dev).eval()
data = bench.make_workload(name, dev)
sec, dist = bench.time_forward(model, data, iters, 3)
seq, ms = bench.time_phases(model, data, iters)
tot = {}
for (ph, arg), t in zip(seq, ms):
    k = bench.PHASE_NAMES[ph] + (f"[{arg}]" if ph == bench._lib.PH_NODE_ENC else "")
    tot[k] = tot.get(k, 0.0) + t
print(f"{name}: forward {sec * 1e3:.3f} ms (median {dist['median']:.3f});", {k: round(v, 3) for k, v in tot.items()})
if os.environ.get("DETAIL"):      # every launch by (phase, arg)
    print("   per launch:", {f"{bench.PHASE_NAMES[ph].split('_kernel')[0]}[{arg}]": round(t, 4) for (ph, arg), t in zip(seq, ms) if t > 2e-3})
