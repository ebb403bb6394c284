import copy, sys, time, types, os
sys.path.insert(0, "/root/repo")
import torch
import mtmc_mpn
from mtmc_mpn import graphs
from oracle import mpn_oracle
torch.manual_seed(0)
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
for (n, e) in [(1000, 40000), (3000, 400000)]:
    d = graphs.random_graph(n, e, seed=5)        # unsorted rows, E/N >= 24
    d.x = torch.nn.functional.normalize(d.x, p=2, dim=0)
    with torch.no_grad():
        ora, oh = mpn_oracle.forward(sd, copy.deepcopy(params), "resnet101", d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
        g = types.SimpleNamespace(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda())
        mm = m.cuda()
        out, h = mm(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): mm(g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(n, e, "err", (out["classified_edges"][0].cpu().double() - ora["classified_edges"][0]).abs().max().item(),
          "h err", ((h.cpu().double() - oh).abs().max() / oh.abs().max()).item(), f"{dt*1e3:.3f} ms")
