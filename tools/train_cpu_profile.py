#!/usr/bin/env python3
"""Host-side profile of the training step (cProfile):  python tools/train_cpu_profile.py"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "5"]
import runpy  # noqa: E402

ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_loop.py"), run_name="not_main")
step = ns["step"]
import torch  # noqa: E402

torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
