#!/usr/bin/env python3
"""Node-encoder column statistics per layer, GPU vs fp64 CPU:  python tools/dbg_enc.py N"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import engine, graphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
torch.manual_seed(0)
params = mtmc_mpn.default_params(num_enc_steps=1, num_class_steps=1)
m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
d = graphs.stress_graph(n, 4 * n, seed=4)
sd = {k: v.detach().double() for k, v in m.state_dict().items()}
a = d.x.double()
ref = []
for l in range(4):
    w, b = sd[f"encoder.node_mlp.fc_layers.{4 * l}.weight"], sd[f"encoder.node_mlp.fc_layers.{4 * l}.bias"]
    y = a @ w.t() + b
    ref.append((y.sum(0), (y * y).sum(0), y.abs().max()))
    g, bt = sd[f"encoder.node_mlp.fc_layers.{4 * l + 1}.weight"], sd[f"encoder.node_mlp.fc_layers.{4 * l + 1}.bias"]
    a = torch.relu((y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5) * g + bt)
m = m.cuda()
eng = engine.ForwardEngine(m)
with torch.no_grad():
    prep = eng.prepare(d.x.cuda(), d.edge_index.cuda(), d.edge_attr.cuda())
    eng.run_phases(prep)
torch.cuda.synchronize()
for l in range(4):
    st = eng.region(prep, "stat_enc_node", l).cpu()
    dim = st.numel() // 2
    s, q = st[:dim], st[dim:]
    es = (s - ref[l][0]).abs().max() / ref[l][1].sqrt().max()
    eq = ((q - ref[l][1]).abs() / ref[l][1]).max()
    print(f"layer {l}: rel err colsum {es:.2e}  colsumsq {eq:.2e}  |Y|max {ref[l][2]:.3f}")
st = eng.region(prep, "stat_enc_node", 0).cpu()
dim = st.numel() // 2
rel = ((st[dim:] - ref[0][1]).abs() / ref[0][1])
print("layer 0 colsumsq rel err by (col % 128) // 32:", [float(rel[(torch.arange(dim) % 128) // 32 == g].max()) for g in range(4)])
print("first 16 cols:", [f"{float(v):.1e}" for v in rel[:16]])
print("worst cols:", torch.topk(rel, 8).indices.tolist())
