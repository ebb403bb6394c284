#!/usr/bin/env python3
"""Forward time over a ladder of graph sizes (camera graphs with 4 equal cameras):  python tools/size_sweep.py [N ...]
Used to pick size-dependent launch shapes (run once per setting of the env knob under test)."""
import copy
import types
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 866, 1400, 2000, 2800, 4000]
dev = torch.device("cuda:0")
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=3)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
out = []
for n in sizes:
    feats = torch.randn(n, 2048, device=dev)
    data = mtmc_mpn.build_graph(feats, [i * 4 // n for i in range(n)])
    ei = data.edge_index
    sec = bench.time_forward(model, data, 100, 10)
    out.append(f"N={n} E={ei.shape[1]}: {sec * 1e6:.1f} us")
print(" | ".join(out))
