#!/usr/bin/env python3
"""bench.run_training_step by itself (wall, GPU-side and host-issue time of a config-3 step):  python tools/train_step_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

r = bench.run_training_step(torch.device("cuda:0"))
print(json.dumps({k: r.get(k) for k in ("ms_per_step", "launch", "ms_per_step_eager", "graph_replay", "step_ms_gpu_side", "step_ms_host_issue")}))
