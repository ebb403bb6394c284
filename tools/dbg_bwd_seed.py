#!/usr/bin/env python3
"""Where does the gradient error of one seed of tests/test_gpu_training.py::test_backward_random_configurations sit?
    python tools/dbg_bwd_seed.py SEED"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
import test_gpu_training as T  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402

seed = int(sys.argv[1])
DROP = len(sys.argv) > 2 and sys.argv[2] == "drop"          # test_dropout_random_configurations instead
g = torch.Generator().manual_seed((3000 if DROP else 2000) + seed)
r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))   # noqa: E731
over = dict(num_enc_steps=r(0, 3), num_class_steps=r(1, 3), node_agg_fn=["sum", "mean", "max"][r(0, 2)],
            reattach_initial_nodes=bool(r(0, 1)), reattach_initial_edges=bool(r(0, 1)))
d = (graphs.camera_graph(tuple(r(8, 30) for _ in range(r(2, 4))), seed=60 + seed) if DROP else
     graphs.camera_graph(tuple(r(4, 20) for _ in range(r(2, 4))), seed=30 + seed))
print(over, "N", d.x.shape[0], "E", d.edge_index.shape[1])
params = mtmc_mpn.default_params(**over) if DROP else T.nodrop(mtmc_mpn.default_params(**over))
torch.manual_seed(0)
m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, T.ARCH)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
labels = (torch.rand(d.edge_index.shape[1], generator=torch.Generator().manual_seed(5)) < 0.15).long()
m = m.cuda().train()
dropout_fn = None
if DROP:
    m.device_seed = torch.tensor([(seed + 1) * 7919], dtype=torch.int64, device="cuda")
    dropout_fn = T.make_dropout_fn((seed + 1) * 7919)
gd = T.to_gpu(d)
out, h = m(gd)
loss = T.loss_of(out, labels.cuda()) + 1e-3 * (h * h).mean()
loss.backward()
ref_loss, ref_grads, _, _, ref_out = T.cpu_autograd(sd, copy.deepcopy(params), d, labels, True, dropout_fn, False)
for a, b in zip(out["classified_edges"], ref_out["classified_edges"]):
    print("forward |dlogit| max", (a.detach().cpu().double() - b.detach()).abs().max().item())
for k, p in m.named_parameters():
    want = ref_grads[k]
    if want is None:
        continue
    err = p.grad.detach().cpu().double() - want
    rel = (err.norm() / max(want.norm().item(), 1e-30)).item()
    line = f"{k:45s} rel L2 {rel:.3e}  max {err.abs().max().item():.3e} / {want.abs().max().item():.3e}"
    if err.dim() == 2 and err.shape[0] >= 32:
        rows = err.norm(dim=1) / want.norm(dim=1).clamp_min(1e-30)
        top = rows.topk(4)
        line += "   worst rows " + ", ".join(f"{int(i)}:{float(v):.2e}" for v, i in zip(top.values, top.indices)) + f"  median row {rows.median().item():.1e}"
    print(line)
# the forward's pre-activations of layer 0 closest to the ReLU kink (fp64 oracle)
