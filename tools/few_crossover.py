#!/usr/bin/env python3
"""Where the few-row encoder kernels (csrc/gemm_few.hip) stop paying against the in-loop / split-K kernels: forward time of sparse
graphs (2 x 20 000 edges: the node encoder dominates) over a ladder of node counts, once per setting of the switch:
    MTMC_FEW_ROWS_MAX=4095 python tools/few_crossover.py        vs        MTMC_GEMM_NO_FEW=1 python tools/few_crossover.py"""
import copy
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [450, 1000, 1500, 2000, 2500, 3000, 4000]
dev = torch.device("cuda:0")
params = mtmc_mpn.default_params(num_enc_steps=1, num_class_steps=1)
torch.manual_seed(0)
model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").to(dev).eval()
out = []
for n in sizes:
    g = graphs.stress_graph(n, 20_000, seed=n)
    data = types.SimpleNamespace(x=g.x.to(dev), edge_index=g.edge_index.to(dev), edge_attr=g.edge_attr.to(dev))
    sec, _ = bench.time_forward(model, data, 60, 10)
    out.append(f"N={n}: {sec * 1e6:.1f} us")
print(("NO_FEW " if os.environ.get("MTMC_GEMM_NO_FEW") else f"FEW<= {os.environ.get('MTMC_FEW_ROWS_MAX', '1024')} ") + " | ".join(out))
