import sys, copy, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import mtmc_mpn
from mtmc_mpn import graphs
from oracle import mpn_oracle
import test_gpu_parity as T
torch.manual_seed(0)
params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
d = graphs.stress_graph(70000, 1000000, seed=4)
with torch.no_grad():
    o64, h64 = mpn_oracle.forward(sd, copy.deepcopy(params), "resnet101", d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
    o32, h32 = mpn_oracle.forward(sd, copy.deepcopy(params), "resnet101", d.x, d.edge_index, d.edge_attr)
    out, h = m.cuda()(T.to_gpu(d))
g = out["classified_edges"][0].cpu().double()
print("f16x3" if not (os.environ.get("MTMC_GEMM_NO_F16") or os.environ.get("MTMC_GEMM_FP32")) else ("bf16x6" if not os.environ.get("MTMC_GEMM_FP32") else "fp32"), "|gpu-fp64| %.2e  |ref32-fp64| %.2e  h rel %.2e  h0-path rel(h32) %.2e" % (
    (g - o64["classified_edges"][0]).abs().max(), (o32["classified_edges"][0].double() - o64["classified_edges"][0]).abs().max(),
    ((h.cpu().double() - h64).abs().max() / h64.abs().max()), ((h32.double() - h64).abs().max() / h64.abs().max())))
