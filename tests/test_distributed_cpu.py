"""Multi-rank path on CPU: the real orchestration code (mtmc_mpn.distributed.ShardedForward: which collective on
which workspace region after which phase) driven over gloo with world_size 2 (and 3, uneven shards), with the CPU
phase backend standing in for the HIP kernels.  Results must equal the single-rank golden vectors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import Case
from mtmc_mpn import distributed as mdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case_name, snap, out_dir, gather=False, own=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_phase_backend import CpuPhaseBackend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = Case(case_name)
        m, d = c.model(), c.graph()
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        n, e = d.x.shape[0], d.edge_index.shape[1]
        lo, hi = mdist.even_ranges(n, world)[rank]
        elo, ehi = mdist.edge_ranges(d.edge_index[0], e, world, snap_to_rows=snap)[rank]
        fwd = mdist.ShardedForward(CpuPhaseBackend(sd, m.spec), m.spec)
        rr = None
        if gather:
            rr = mdist.row_ranges_of(d.edge_index[:, elo:ehi])
            assert (rr is not None) == snap, "row-disjoint shards expected exactly for snapped boundaries"
        if own:                                          # encode the rows the rank projects: no h0 exchange
            lo, hi = mdist.tile_rows(rr, n)[rank]
        with torch.no_grad():
            logits, h = fwd(d.x[lo:hi], (lo, hi, n), d.edge_index[:, elo:ehi], d.edge_attr[elo:ehi], e, rr, own)
        torch.save({"elo": elo, "ehi": ehi, "logits": [l.clone() for l in logits], "h": h.clone()},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run(case_name, world, snap, tmp_path, gather=False, own=False):
    mp.spawn(_worker, args=(world, _free_port(), case_name, snap, str(tmp_path), gather, own), nprocs=world, join=True)
    c = Case(case_name)
    parts = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    assert parts[0]["elo"] == 0 and parts[-1]["ehi"] == c.meta["E"]
    for i in range(c.meta["n_out"]):
        full = torch.cat([p["logits"][i] for p in parts], 0)
        assert full.shape[0] == c.meta["E"]
        assert (full[c.sub_idx] - c.logits(i)).abs().max().item() <= 1e-4
        assert (full[c.sub_idx].double() - c.logits(i, f64=True)).abs().max().item() <= 5e-5
    scale = max(1.0, c.h(f64=True).abs().max().item())
    for p in parts:                                 # h is replicated: every rank must hold the full result
        assert (p["h"].double() - c.h(f64=True)).abs().max().item() <= 1e-4 * scale
    return parts


def test_single_rank_cpu_backend_matches_golden():
    """The stand-in backend itself, phase by phase without any collective, reproduces the reference."""
    from cpu_phase_backend import CpuPhaseBackend
    for name in ("g2_random_L3_C3", "g5_mean", "g5_max", "g5_reattach_both_s02", "g5_L0", "g3_cams324_L2"):
        c = Case(name)
        m, d = c.model(), c.graph()
        be = CpuPhaseBackend({k: v.detach() for k, v in m.state_dict().items()}, m.spec)
        with torch.no_grad():
            ctx = be.prepare(d.x, d.edge_index, d.edge_attr)
            be.run_phase_list(ctx, be.phase_list())
            logits, h = be.outputs(ctx)
        for i, lg in enumerate(logits):
            assert (lg[c.sub_idx] - c.logits(i)).abs().max().item() <= 1e-4, name
        assert (h.double() - c.h(f64=True)).abs().max().item() <= 1e-4 * max(1.0, c.h(f64=True).abs().max().item())


@pytest.mark.parametrize("name,world,snap", [
    ("g2_random_L3_C3", 2, False),          # unsorted rows, rows straddle the shard boundary
    ("g5_mean", 2, False),                  # needs the global degree
    ("g5_max", 2, False),                   # all-reduce MAX of node states
    ("g5_reattach_both_s02", 2, True),      # row-sorted, boundaries snapped to row changes
    ("g3_cams324_L2", 3, False),            # N = 9 over 3 ranks (even), tiny shards
    ("g1_random_L1", 3, False),             # N = 64 over 3 ranks: uneven node shards -> broadcast path
])
def test_sharded_forward_over_gloo(name, world, snap, tmp_path):
    _run(name, world, snap, tmp_path)


@pytest.mark.parametrize("name,world,snap", [
    ("g5_reattach_both_s02", 2, True),      # row-sorted + snapped: complete rows per rank -> all-gather of node states
    ("g4_s02_L3", 3, True),                 # three ranks, uneven row ranges (padded gather)
    ("g5_max", 2, False),                   # unsorted rows: row_ranges_of() says no, falls back to all-reduce
])
def test_row_complete_shards_use_all_gather(name, world, snap, tmp_path):
    _run(name, world, snap, tmp_path, gather=True)


@pytest.mark.parametrize("name,world", [("g5_reattach_both_s02", 2), ("g4_s02_L3", 3)])
def test_row_complete_shards_that_encode_their_own_rows(name, world, tmp_path):
    """own_rows: the encoder's row partition is the edge shards' source-row partition; h0 is never exchanged."""
    _run(name, world, True, tmp_path, gather=True, own=True)


def test_tile_rows():
    assert mdist.tile_rows([(2, 5), (7, 9)], 12) == [(0, 7), (7, 12)]          # gaps go to the earlier rank
    assert mdist.tile_rows([(6, 9), (0, 0), (1, 4)], 10) == [(6, 10), (0, 0), (0, 6)]   # rank order is free; empty shard
    assert mdist.tile_rows([(0, 0), (0, 0)], 5) == [(0, 5), (0, 0)]            # no edges at all


def _gap_worker(rank, world, port, out_dir, own, L):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_phase_backend import CpuPhaseBackend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, d = _gap_case(L)
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        n, e = d.x.shape[0], d.edge_index.shape[1]
        lo, hi = mdist.even_ranges(n, world)[rank]
        elo, ehi = mdist.edge_ranges(d.edge_index[0], e, world, snap_to_rows=True)[rank]
        rr = mdist.row_ranges_of(d.edge_index[:, elo:ehi])
        assert rr is not None
        if own:
            lo, hi = mdist.tile_rows(rr, n)[rank]
        with torch.no_grad():
            logits, h = mdist.ShardedForward(CpuPhaseBackend(sd, m.spec), m.spec)(
                d.x[lo:hi], (lo, hi, n), d.edge_index[:, elo:ehi], d.edge_attr[elo:ehi], e, rr, own)
            # the same without the final all-gather: the rank's own rows must already be final
            _, h_own = mdist.ShardedForward(CpuPhaseBackend(sd, m.spec), m.spec)(
                d.x[lo:hi], (lo, hi, n), d.edge_index[:, elo:ehi], d.edge_attr[elo:ehi], e, rr, own, replicate_h=False)
        tlo, thi = mdist.tile_rows(rr, n)[rank]
        assert torch.equal(h_own[tlo:thi], h[tlo:thi])
        torch.save({"logits": [l.clone() for l in logits], "h": h.clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _gap_case(L=2):
    """Row-sorted camera graph from which every out-edge of some nodes is removed (they stay as edge targets): rows
    no rank's edges start from -- at the front, between two shards and at the end of the node range."""
    import copy
    import mtmc_mpn
    from mtmc_mpn import graphs
    d = graphs.camera_graph((9, 7, 8), seed=4)
    n = d.x.shape[0]
    dead = torch.tensor([0, 1, 11, 12, 13, n - 1])
    keep = ~torch.isin(d.edge_index[0], dead)
    d.edge_index, d.edge_attr = d.edge_index[:, keep].contiguous(), d.edge_attr[keep].contiguous()
    torch.manual_seed(3)
    p = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=2)
    p["node_agg_fn"] = "mean"
    return mtmc_mpn.MOTMPNet(copy.deepcopy(p), None, "resnet101").eval(), d


@pytest.mark.parametrize("world,own,L", [(2, False, 2), (3, False, 2), (3, True, 2), (2, True, 0)])
def test_rows_without_out_edges_in_row_complete_shards(world, own, L, tmp_path):
    from cpu_phase_backend import CpuPhaseBackend
    mp.spawn(_gap_worker, args=(world, _free_port(), str(tmp_path), own, L), nprocs=world, join=True)
    m, d = _gap_case(L)
    be = CpuPhaseBackend({k: v.detach() for k, v in m.state_dict().items()}, m.spec)
    with torch.no_grad():
        ctx = be.prepare(d.x, d.edge_index, d.edge_attr)
        be.run_phase_list(ctx, be.phase_list())
        want_logits, want_h = be.outputs(ctx)
    parts = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    for i, w in enumerate(want_logits):
        got = torch.cat([q["logits"][i] for q in parts], 0)
        assert got.shape == w.shape and (got - w).abs().max().item() <= 1e-5
    for q in parts:
        assert torch.isfinite(q["h"]).all() and (q["h"] - want_h).abs().max().item() <= 1e-5
    if L > 0:
        assert (want_h[0] == 0).all()               # a node nothing flows out of aggregates nothing


def test_ranges():
    assert mdist.even_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    row = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2, 2, 3])
    r = mdist.edge_ranges(row, 10, 2, snap_to_rows=True)
    assert r == [(0, 5), (5, 10)] and mdist.edge_ranges(row, 10, 2) == [(0, 5), (5, 10)]
    r3 = mdist.edge_ranges(row, 10, 3, snap_to_rows=True)
    assert r3[0][0] == 0 and r3[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(r3, r3[1:]))
    for lo, hi in r3[:-1]:
        assert hi == 10 or row[hi] != row[hi - 1]


def test_step_plan_collective_count():
    """<= 14 collectives per L = 3 forward on row-complete shards (was one per phase, ~30), and
    one library call per collective (everything between two exchanges runs back to back)."""
    import copy
    import mtmc_mpn
    from mtmc_mpn import _lib
    spec = mtmc_mpn.MOTMPNet(copy.deepcopy(mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)), None, "resnet101").spec
    own = mdist.step_plan(spec, local_rows=True, own_rows=True)
    exch = [w for _, (w, _) in own if w is not None]
    assert len(exch) == 13 and len(own) == 14          # + the final all-gather of the node state = 14 collectives
    assert [w for w in exch[:4]] == ["enc_merged", "enc_merged", "stat_enc_node", "stat_enc_node"]
    assert exch[4:] == ["Pc", "round_z1", "round_m_z2"] * 3
    # pass C of round r and the projection of round r + 1 share a library call; END rides with the last pass C
    assert own[7][0] == [(_lib.PH_ROUND_C, 0), (_lib.PH_ROUND_PROJ, 1)] and own[-1][0] == [(_lib.PH_ROUND_C, 2), (_lib.PH_END, 0)]
    # every phase of the single-GPU sequence appears exactly once, dependencies in order
    flat = [pa for phases, _ in own for pa in phases]
    assert sorted(flat) == sorted([(_lib.PH_BEGIN, 0), (_lib.PH_EDGE_ENC, 0), (_lib.PH_NODE_H0, 0), (_lib.PH_END, 0)] +
                                  [(p, l) for l in range(4) for p in (_lib.PH_NODE_ENC, _lib.PH_NODE_COMBINE)] +
                                  [(p, r) for r in range(3) for p in (_lib.PH_ROUND_PROJ, _lib.PH_ROUND_A, _lib.PH_ROUND_B,
                                                                      _lib.PH_ROUND_STAT, _lib.PH_ROUND_C)])
    assert flat.index((_lib.PH_BEGIN, 0)) < flat.index((_lib.PH_NODE_ENC, 0)) < flat.index((_lib.PH_EDGE_ENC, 0))
    gen = mdist.step_plan(spec, local_rows=False, own_rows=False)
    assert [w for _, (w, _) in gen if w is not None].count("agg") == 3 and ("h0", 0) in [x for _, x in gen]


def test_merged_statistics_blocks_are_adjacent_in_the_workspace():
    """The one-message all-reduces rely on the layout: stat_attr | encoder layer 0, stat_enc2 | encoder layer 1."""
    import copy
    import ctypes
    import mtmc_mpn
    from mtmc_mpn import _lib, engine
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(mtmc_mpn.default_params(num_enc_steps=3)), None, "resnet101")
    eng = engine.ForwardEngine(m)
    model = eng.model_struct(next(m.parameters()).device)
    lay = _lib.WsLayout()
    assert eng.lib.mtmc_mpn_workspace_layout(ctypes.byref(model), 450, 150454, ctypes.byref(lay)) == 0
    blk = 8 * _lib.STAT_REPLICAS
    assert lay.stat_attr_off + blk * _lib.ATTR_STRIDE == lay.stat_enc_layer_off[0]
    assert lay.stat_enc_layer_off[0] + 16 * 1024 == lay.stat_enc2_off
    assert lay.stat_enc2_off + blk * _lib.ENC2_STRIDE == lay.stat_enc_layer_off[1]
    assert lay.stat_enc_layer_off[1] + 16 * 512 == lay.stat_enc_layer_off[2] < lay.stat_enc_layer_off[3] < lay.stat_round_off
    assert lay.stat_round_off < lay.zero_bytes
