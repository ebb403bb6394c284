"""Data-parallel training: torch's DistributedDataParallel around the drop-in module, one graph per rank (the
reference trains one graph per iteration, train.py:316-356), gradients averaged by DDP.  Two ranks share the one
GPU of the test box over gloo; on a multi-GPU node the same code runs over RCCL."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _graph(seed):
    from mtmc_mpn import graphs
    return graphs.camera_graph((13 + seed, 9, 11), seed=20 + seed)


def _loss(model, d, dev):
    import types
    import mtmc_mpn
    g = types.SimpleNamespace(x=d.x.to(dev), edge_index=d.edge_index.to(dev), edge_attr=d.edge_attr.to(dev))
    labels = (torch.rand(d.edge_index.shape[1], generator=torch.Generator().manual_seed(3)) < 0.2).long().to(dev)
    out, h = model(g)
    return sum(mtmc_mpn.cross_entropy(o, labels) for o in out["classified_edges"]) + 1e-3 * (h * h).mean()


def _params():
    import mtmc_mpn
    p = mtmc_mpn.default_params(num_enc_steps=2, num_class_steps=2)
    p["encoder_feats_dict"]["nodes"]["resnet101"]["dropout_p"] = 0.0
    p["edge_model_feats_dict"]["dropout_p"] = 0.0
    p["node_model_feats_dict"]["dropout_p"] = 0.0
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mtmc_mpn
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = mtmc_mpn.MOTMPNet(copy.deepcopy(_params()), None, "resnet101").to(dev).train()
        ddp = torch.nn.parallel.DistributedDataParallel(model)
        loss = _loss(ddp, _graph(rank), dev)
        loss.backward()
        torch.cuda.synchronize()
        torch.save({k: p.grad.cpu() for k, p in model.named_parameters()}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ddp_averages_the_gradients_of_the_ranks(tmp_path):
    import mtmc_mpn
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2)]
    # single-process reference: the mean of the two graphs' gradients
    dev = torch.device("cuda:0")
    want = None
    for r in range(2):
        torch.manual_seed(0)
        model = mtmc_mpn.MOTMPNet(copy.deepcopy(_params()), None, "resnet101").to(dev).train()
        _loss(model, _graph(r), dev).backward()
        grads = {k: p.grad.cpu() for k, p in model.named_parameters()}
        want = grads if want is None else {k: (want[k] + grads[k]) / 2 for k in grads}
    for k in want:
        scale = max(1e-6, want[k].abs().max().item())
        for r in range(2):
            assert (got[r][k] - want[k]).abs().max().item() <= 1e-5 * scale + 1e-7, k
