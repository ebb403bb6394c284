#!/usr/bin/env python3
"""Training step with the per-step loss vs cross_entropy_steps, same process:  python tools/ce_ab.py"""
import os
import runpy
import sys
import time

sys.argv = [sys.argv[0], "5"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_loop.py"), run_name="not_main")
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402

model, opt, data, labels = ns["model"], ns["opt"], ns["data"], ns["labels"]


def step(fused):
    opt.zero_grad(set_to_none=True)
    out, _ = model(data)
    steps = out["classified_edges"]
    loss = mtmc_mpn.cross_entropy_steps(steps, labels) if fused else sum(mtmc_mpn.cross_entropy(o, labels) for o in steps)
    loss.backward()
    opt.step()


for rep in range(3):
    for fused in (False, True):
        for _ in range(20):
            step(fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            step(fused)
        torch.cuda.synchronize()
        print(f"{'one pass over the steps' if fused else 'per-step loss':24s} {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms/step", flush=True)
