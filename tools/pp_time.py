#!/usr/bin/env python3
"""Post-processing timing on one GPU:  python tools/pp_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

print(json.dumps(bench.run_postprocess(torch.device("cuda:0"), with_cpu="--cpu" in sys.argv), indent=1))
