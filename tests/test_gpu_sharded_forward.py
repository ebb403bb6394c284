"""Edge-partitioned forward on the REAL kernels: 2 and 3 ranks share the test box's one GPU over gloo (on a multi-GPU
node the same code runs over RCCL, one rank per GPU), every rank encodes its node rows and walks its edge slice, and
the stitched logits / replicated node states must equal the one-process forward of the same graph.  Covers both
exchange schemes (all-reduce of the node state; row-complete shards that keep their rows' state and all-gather the
column projections), all aggregations, and node rows no edge starts from."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _case(agg):
    import mtmc_mpn
    from mtmc_mpn import graphs
    if agg.endswith("+big"):       # 5 003 nodes / 240k edges, random sorted graph (some rows empty): hundreds of workgroups
        agg = agg[:-4]             # per node kernel and row ranges that start anywhere
        d = graphs.stress_graph(5003, 120000, seed=8)
        p = mtmc_mpn.default_params(num_enc_steps=2, num_class_steps=1)
        p["node_agg_fn"] = agg
        return d, p
    d = graphs.camera_graph((31, 24, 17, 29), seed=5)
    if agg.endswith("+gaps"):      # nodes without out-edges (still edge targets): at the front, inside, at the end
        agg = agg[:-5]
        dead = torch.tensor([0, 1, 2, 40, 41, 57, d.x.shape[0] - 2, d.x.shape[0] - 1])
        keep = ~torch.isin(d.edge_index[0], dead)
        d.edge_index, d.edge_attr = d.edge_index[:, keep].contiguous(), d.edge_attr[keep].contiguous()
    p = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=2)
    p["node_agg_fn"] = agg
    if agg == "max":
        p["reattach_initial_nodes"] = True
    return d, p


def _worker(rank, world, port, agg, snap, out_dir, own=False, backend="gloo", alone=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mtmc_mpn
    from mtmc_mpn import distributed as mdist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # gloo: all ranks share cuda:0 (the 1-GPU test box);  nccl (= RCCL): one GPU per rank, as bench.py --gpus N runs it
    dev = torch.device(f"cuda:{rank}" if backend == "nccl" else "cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, p = _case(agg)
        torch.manual_seed(0)
        model = mtmc_mpn.MOTMPNet(copy.deepcopy(p), None, "resnet101").to(dev).eval()
        n, e = d.x.shape[0], d.edge_index.shape[1]
        lo, hi = mdist.even_ranges(n, world)[rank]
        elo, ehi = mdist.edge_ranges(d.edge_index[0], e, world, snap_to_rows=snap)[rank]
        ei = d.edge_index[:, elo:ehi].contiguous().to(dev)
        rr = mdist.row_ranges_of(ei) if snap else None
        if snap:
            assert rr is not None
        if own:
            lo, hi = mdist.tile_rows(rr, n)[rank]
        with torch.no_grad():
            out, h = mdist.sharded_forward(model, d.x[lo:hi].contiguous().to(dev), (lo, hi, n), ei,
                                           d.edge_attr[elo:ehi].contiguous().to(dev), e, row_ranges=rr, own_rows=own,
                                           exchange_alone=alone)
        torch.cuda.synchronize()
        torch.save({"logits": [o.cpu() for o in out["classified_edges"]], "h": h.cpu(), "edges": (elo, ehi)},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,agg,snap,own", [(2, "sum", False, False), (2, "mean", True, False), (3, "max", False, False),
                                                (3, "sum", True, False), (3, "mean+gaps", True, False),
                                                (2, "max", True, True), (3, "sum+gaps", True, True),
                                                (3, "mean+big", True, True), (2, "sum+big", False, False)])
def test_sharded_forward_on_the_gpu_kernels(world, agg, snap, own, tmp_path):
    _run_and_compare(world, agg, snap, own, tmp_path, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL path at world size 2 needs two GPUs (the driver's 8-GPU node)")
@pytest.mark.parametrize("agg,snap,own", [("sum", False, False), ("mean", True, False), ("max", True, True),
                                          ("sum+gaps", True, True), ("mean+big", True, True), ("sum+big", False, False)])
def test_sharded_forward_over_rccl_two_gpus(agg, snap, own, tmp_path):
    """The `nccl` branches of distributed.py (all_gather_into_tensor of Pc / h0 / the final h, device-side row_ranges_of)
    with one GPU per rank: general shards, row-complete shards and own_rows, against the single-GPU forward."""
    _run_and_compare(2, agg, snap, own, tmp_path, "nccl")


@pytest.mark.parametrize("agg,snap,own", [("sum", False, False), ("mean", True, False), ("max", True, True),
                                          ("sum+gaps", True, True)])
def test_rccl_calls_rehearsed_in_a_group_of_one(agg, snap, own, tmp_path):
    """Every RCCL call of distributed.py (all_reduce of the statistics blocks and of the node state, all_gather_into_tensor
    of Pc / h0 / the final h, the device-side all-gather of row_ranges_of) issued for real on the one GPU this box has:
    a group of one rank moves nothing, but dtypes, shapes, contiguity and stream order are what RCCL sees at any size."""
    _run_and_compare(1, agg, snap, own, tmp_path, "nccl", alone=True)


def _run_and_compare(world, agg, snap, own, tmp_path, backend, alone=False):
    import types
    import mtmc_mpn
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, agg, snap, str(tmp_path), own, backend, alone), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    dev = torch.device("cuda:0")
    d, p = _case(agg)
    torch.manual_seed(0)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(p), None, "resnet101").to(dev).eval()
    with torch.no_grad():
        out, h = model(types.SimpleNamespace(x=d.x.to(dev), edge_index=d.edge_index.to(dev), edge_attr=d.edge_attr.to(dev)))
    assert [q["edges"] for q in parts][0][0] == 0 and parts[-1]["edges"][1] == d.edge_index.shape[1]
    for step, want in enumerate(out["classified_edges"]):
        got = torch.cat([q["logits"][step] for q in parts])
        assert got.shape == want.shape
        assert (got - want.cpu()).abs().max().item() < 1e-4          # the parity tolerance; measured ~1e-6
    for q in parts:                                                    # node states replicated on every rank
        assert (q["h"] - h.cpu()).abs().max().item() < 1e-4 * max(1.0, h.abs().max().item())
