#!/usr/bin/env python3
"""One-off fuzz of the many-row / many-edge dispatch regimes against the fp64 oracle (not a test: minutes of CPU oracle time).
   python tools/fuzz_large.py [cases] [seed]"""
import copy
import os
import random
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import engine, graphs  # noqa: E402
from oracle import mpn_oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for c in range(cases):
    n = rng.choice([4097, 5000, 8191, 12000, 20011, 33000, 49152, 50001])
    deg = rng.choice([26, 40, 70, 130])
    pairs = max(270_000, n * deg // 2)
    pairs = min(pairs, 1_500_000)
    L = rng.choice([1, 2, 3])
    Cs = rng.choice([1, L])
    over = rng.choice([{}, {}, dict(node_agg_fn="mean"), dict(reattach_initial_nodes=True), dict(reattach_initial_edges=True)])
    det = rng.random() < 0.3
    d = graphs.stress_graph(n, pairs, seed=100 + c)
    drop = rng.choice([0, 0, 1, 17, 63])
    if drop:
        E0 = d.edge_index.shape[1]
        d = types.SimpleNamespace(x=d.x, edge_index=d.edge_index[:, :E0 - drop].contiguous(), edge_attr=d.edge_attr[:E0 - drop].contiguous())
    E = d.edge_index.shape[1]
    torch.manual_seed(c)
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=Cs, **over)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    m.deterministic = det
    plan = engine.ForwardEngine(m).plan(n, E, flags=1 if det else 0)
    with torch.no_grad():
        out, h = m(types.SimpleNamespace(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda()))
        want, h64 = mpn_oracle.forward(sd, copy.deepcopy(params), "resnet101", d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
    errs = [(g.cpu().double() - w).abs().max().item() for g, w in zip(out["classified_edges"], want["classified_edges"])]
    herr = (h.cpu().double() - h64).abs().max().item() / max(1.0, h64.abs().max().item())
    worst = max(worst, max(errs))
    print(f"case {c}: N={n} E={E} (E%64={E % 64}) L={L} Cs={Cs} {over} det={det} enc={plan.enc_kernel} pass_c={plan.pass_c}: "
          f"max |dlogit| {max(errs):.2e}, h rel {herr:.2e}", flush=True)
    assert max(errs) <= 1e-4 and herr <= 1e-4
print(f"all {cases} cases within 1e-4 (worst {worst:.2e})")
