#!/usr/bin/env python3
"""Per-phase HIP-event timings of one forward (development aid):  python tools/phase_times.py s02 cfg4 ..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for name in (sys.argv[1:] or ["s02"]):
        big = name in ("cfg4", "cfg5")
        r = bench.run_single(name, dev, 10 if big else 200, 3 if big else 20, with_cpu=False, phase_iters=5 if big else 50)
        print(name, f"{r['ms_per_step']*1e3:.1f} us/fwd  {r['value']/1e6:.1f} M edges/s  sum(phases) {r['phase_ms_sum']*1e3:.1f} us")
        print("   ", json.dumps({k: round(v * 1e3, 1) for k, v in r["phase_ms"].items()}))
        print("   ", json.dumps(r["roofline"]))
