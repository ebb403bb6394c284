"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the drop-in module
and hence the C ABI, against (i) the golden vectors generated from the reference and (ii) the CPU oracle
evaluated on the same seeded inputs.

Tolerances (BASELINE.json north_star / SURVEY.md 8(c)):  max |logit_gpu - logit_ref| <= 1e-4 in fp32;
predicted edge labels equal on every edge whose reference margin |logit1 - logit0| exceeds 2e-4
(the fp32 reference itself flips labels inside that band when the edge order changes).
"""
import copy
import types

import pytest
import torch

import mtmc_mpn
from golden_util import ARCH, Case, case_names, sha
from mtmc_mpn import graphs

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
MARGIN_GUARD = 2e-4
FWD_CASES = case_names()


def to_gpu(d):
    # edge_index keeps the callers' non-contiguous [E,2].T layout on the device (inference.py:413)
    ei = d.edge_index
    ei_gpu = ei.t().contiguous().cuda().t() if not ei.is_contiguous() else ei.cuda()
    return types.SimpleNamespace(x=d.x.cuda(), edge_index=ei_gpu, edge_attr=d.edge_attr.cuda())


def label_mismatches(got, ref64):
    margin = (ref64[:, 1] - ref64[:, 0])
    guard = margin.abs() > MARGIN_GUARD
    pred_ref = margin > 0
    pred_got = (got[:, 1] - got[:, 0]) > 0
    return int(((pred_ref != pred_got) & guard).sum()), int((~guard).sum())


def test_only_one_hip_runtime_loaded():
    """libmtmc_mpn.so must bind to the HIP runtime torch already loaded (one runtime, one context)."""
    from mtmc_mpn import _lib
    _lib.load()
    torch.zeros(1, device="cuda")
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, libs


@pytest.mark.parametrize("name", FWD_CASES)
def test_forward_matches_golden_and_oracle(name):
    from oracle import mpn_oracle
    c = Case(name)
    m, d = c.model(), c.graph()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    with torch.no_grad():
        out, h = m(to_gpu(d))
        ora, oh = mpn_oracle.forward(sd, c.params(), ARCH, d.x, d.edge_index, d.edge_attr)
        ora64, oh64 = mpn_oracle.forward(sd, c.params(), ARCH, d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
    logits = out["classified_edges"]
    assert isinstance(out, dict) and len(logits) == c.meta["n_out"] == len(ora["classified_edges"])
    assert h.shape == (c.meta["N"], 32) and h.dtype == torch.float32 and h.is_cuda
    # x and edge_index are seeded integers / randn: identical on every host.  edge_attr (and normalize(x)) may differ from
    # the reference run in the last bit when this host's CPU takes another vector path -- irrelevant at 1e-4, so the
    # golden comparison below is unconditional (round 1 skipped it in that case).
    assert sha(d.edge_index) == c.meta["input_sha"]["edge_index"]
    for i, lg in enumerate(logits):
        assert lg.shape == (c.meta["E"], 2) and lg.dtype == torch.float32 and lg.is_cuda
        got = lg.cpu()
        err_oracle = (got - ora["classified_edges"][i]).abs().max().item()
        assert err_oracle <= LOGIT_TOL, f"{name}[{i}] vs CPU oracle: {err_oracle:.3e}"
        err_gold = (got[c.sub_idx] - c.logits(i)).abs().max().item()
        assert err_gold <= LOGIT_TOL, f"{name}[{i}] vs golden (reference): {err_gold:.3e}"
        err64 = (got.double() - ora64["classified_edges"][i]).abs().max().item()
        ref_err64 = (ora["classified_edges"][i].double() - ora64["classified_edges"][i]).abs().max().item()
        assert err64 <= max(4 * ref_err64, 2e-5), f"{name}[{i}]: |gpu-fp64| {err64:.2e} vs reference's own {ref_err64:.2e}"
        bad, unguarded = label_mismatches(got.double(), ora64["classified_edges"][i])
        assert bad == 0, f"{name}[{i}]: {bad} label flips outside the margin guard ({unguarded} edges inside it)"
    scale = max(1.0, oh64.abs().max().item())
    assert (h.cpu().double() - oh64).abs().max().item() <= 1e-4 * scale


def test_edge_permutation_invariance():
    """G7: permuting the edge list permutes the logits and leaves h unchanged (BatchNorm couples all edges)."""
    a, b = Case("g4_s02_L3"), Case("g7_s02_L3_perm")
    m = a.model().cuda().eval()
    da, db = a.graph(), b.graph()
    perm = torch.randperm(da.edge_index.shape[1], generator=torch.Generator().manual_seed(b.meta["perm_seed"]))
    with torch.no_grad():
        oa, ha = m(to_gpu(da))
        ob, hb = m(to_gpu(db))
    la, lb = oa["classified_edges"][-1].cpu(), ob["classified_edges"][-1].cpu()
    assert (la[perm] - lb).abs().max().item() <= 2e-5
    assert (ha - hb).abs().max().item() <= 1e-5 * max(1.0, ha.abs().max().item())


def test_contiguous_and_transposed_edge_index_agree():
    c = Case("g4_s02_L1")
    m, d = c.model().cuda().eval(), c.graph()
    g1 = to_gpu(d)
    g2 = types.SimpleNamespace(x=g1.x, edge_index=g1.edge_index.contiguous(), edge_attr=g1.edge_attr)
    assert not g1.edge_index.is_contiguous() and g2.edge_index.is_contiguous()
    with torch.no_grad():
        o1, h1 = m(g1)
        o2, h2 = m(g2)
    assert torch.equal(o1["classified_edges"][0], o2["classified_edges"][0])


def test_repeatability():
    """Same call twice: BatchNorm statistics are fp64 (order-insensitive to the last fp32 bit); the
    segment sums use fp32 atomics in this build, so h / logits may move in the last bits only."""
    c = Case("g4_s02_L3")
    m, g = c.model().cuda().eval(), to_gpu(c.graph())
    with torch.no_grad():
        o1, h1 = m(g)
        o2, h2 = m(g)
    assert (o1["classified_edges"][0] - o2["classified_edges"][0]).abs().max().item() <= 2e-6
    assert (h1 - h2).abs().max().item() <= 1e-6 * h1.abs().max().item()


def test_error_behaviour():
    c = Case("g3_cams324_L2")
    m, d = c.model(), c.graph()
    with pytest.raises(RuntimeError):          # CPU module + CPU tensors: refused, no fallback
        m(d)
    m = m.cuda().eval()
    g = to_gpu(d)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            m(types.SimpleNamespace(x=g.x[:, :100], edge_index=g.edge_index, edge_attr=g.edge_attr))
        with pytest.raises(RuntimeError):
            m(types.SimpleNamespace(x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr[:-1]))
        with pytest.raises(ValueError):        # BatchNorm over a single row, as the reference (models/mlp.py:16)
            m(types.SimpleNamespace(x=g.x[:1], edge_index=g.edge_index * 0, edge_attr=g.edge_attr))
        with pytest.raises(ValueError):
            m(types.SimpleNamespace(x=g.x, edge_index=g.edge_index[:, :1], edge_attr=g.edge_attr[:1]))
        m.check_indices = True
        bad = g.edge_index.clone()
        bad[1, 3] = 10 ** 6
        with pytest.raises(IndexError):
            m(types.SimpleNamespace(x=g.x, edge_index=bad, edge_attr=g.edge_attr))
        m.check_indices = False
        out, h = m(g)                           # still healthy afterwards
        assert torch.isfinite(out["classified_edges"][-1]).all()


def test_module_protocol():
    """The caller-side protocol of SURVEY.md 8(b): .cuda(), .eval(), state_dict round trip, parameters()."""
    c = Case("g1_random_L1")
    m = c.model().cuda().eval()
    g = to_gpu(c.graph())
    with torch.no_grad():
        ref, _ = m(g)
    m2 = mtmc_mpn.MOTMPNet(c.params(), None, ARCH).cuda().eval()
    m2.load_state_dict(copy.deepcopy(m.state_dict()), strict=True)
    with torch.no_grad():
        got, _ = m2(g)
    assert torch.equal(ref["classified_edges"][0], got["classified_edges"][0])
    assert len(list(m.parameters())) == 34


@pytest.mark.parametrize("fn", ["sum", "mean", "max"])
def test_scatter_surface(fn):
    from oracle import mpn_oracle
    from mtmc_mpn import ops
    g = torch.Generator().manual_seed(11)
    src = torch.randn(5000, 32, generator=g)
    idx = torch.randint(0, 300, (5000,), generator=g)
    idx[idx == 7] = 8                                  # leave an empty row
    want = mpn_oracle.aggregate(src, idx, 301, fn)
    if fn == "sum":
        got = ops.scatter_add(src.cuda(), idx.cuda(), dim=0, dim_size=301)
    elif fn == "mean":
        got = ops.scatter_mean(src.cuda(), idx.cuda(), dim=0, dim_size=301)
    else:
        got, arg = ops.scatter_max(src.cuda(), idx.cuda(), dim=0, dim_size=301)
        assert torch.equal(got.cpu(), want)            # max is exact
        a = arg.cpu()
        assert (a[7] == 5000).all() and (a[300] == 5000).all()
        rows = torch.arange(301)[:, None].expand_as(a)
        ok = a < 5000
        assert torch.equal(src[a[ok], torch.arange(32).expand_as(a)[ok]], want[ok])
        assert (idx[a[ok]] == rows[ok]).all()
    assert got.shape == (301, 32)
    assert (got.cpu() - want).abs().max().item() <= 1e-4
    assert (got[7] == 0).all() and (got[300] == 0).all()


def test_standalone_mlp_forward():
    from oracle import mpn_oracle
    c = Case("g1_random_L1")
    m, d = c.model(), c.graph()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    plans = mpn_oracle.model_plans(c.params(), ARCH)
    want_nodes = mpn_oracle.mlp_forward(d.x, sd, "encoder.node_mlp", plans["enc_node"])
    want_edges = mpn_oracle.mlp_forward(d.edge_attr, sd, "encoder.edge_mlp", plans["enc_edge"])
    m = m.cuda().eval()
    with torch.no_grad():
        e_out, n_out = m.encoder(d.edge_attr.cuda(), d.x.cuda())
    assert (n_out.cpu() - want_nodes).abs().max().item() <= 2e-5
    assert (e_out.cpu() - want_edges).abs().max().item() <= 2e-5


def test_mid_size_sorted_graph_against_oracle():
    """A 20k-node / 2M-edge stress graph (config-4 recipe, scaled so the CPU oracle takes seconds)."""
    from oracle import mpn_oracle
    torch.manual_seed(0)
    params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    d = graphs.stress_graph(20000, 1000000, seed=4)
    with torch.no_grad():
        ora64, oh64 = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
        out, h = m.cuda()(to_gpu(d))
    got = out["classified_edges"][0].cpu().double()
    assert (got - ora64["classified_edges"][0]).abs().max().item() <= LOGIT_TOL
    bad, _ = label_mismatches(got, ora64["classified_edges"][0])
    assert bad == 0
    assert (h.cpu().double() - oh64).abs().max().item() <= 1e-4 * max(1.0, oh64.abs().max().item())


@pytest.mark.parametrize("n,e", [(1000, 40000), (900, 33000)])
def test_unsorted_list_with_many_edges_per_node(n, e):
    """Few-edge lists with >= 24 edges per node take the matrix-core pass C alone, whatever their order: on a randomly
    ordered list every 32-edge group touches ~32 rows -- the per-edge fallback of pass_c_mfma_kernel -- and duplicates /
    self-loops occur.  Against the fp64 oracle."""
    from oracle import mpn_oracle
    torch.manual_seed(0)
    params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=2)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    d = graphs.random_graph(n, e, seed=5)
    d.x = torch.nn.functional.normalize(d.x, p=2, dim=0)
    with torch.no_grad():
        ora64, oh64 = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
        out, h = m.cuda()(to_gpu(d))
    for i in range(2):
        got = out["classified_edges"][i].cpu().double()
        assert (got - ora64["classified_edges"][i]).abs().max().item() <= LOGIT_TOL
        bad, _ = label_mismatches(got, ora64["classified_edges"][i])
        assert bad == 0
    assert (h.cpu().double() - oh64).abs().max().item() <= 1e-4 * max(1.0, oh64.abs().max().item())


def _gpu_shard_worker(rank, world, port, case_name, out_dir):
    import os
    import torch.distributed as dist
    from mtmc_mpn import distributed as mdist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share the one GPU: gloo moves the data
    try:
        c = Case(case_name)
        m, d = c.model().cuda().eval(), c.graph()
        n, e = d.x.shape[0], d.edge_index.shape[1]
        lo, hi = mdist.even_ranges(n, world)[rank]
        elo, ehi = mdist.edge_ranges(d.edge_index[0], e, world)[rank]
        g = to_gpu(d)
        with torch.no_grad():
            out, h = mdist.sharded_forward(m, g.x[lo:hi], (lo, hi, n), g.edge_index[:, elo:ehi],
                                           g.edge_attr[elo:ehi].contiguous(), e)
        torch.cuda.synchronize()
        torch.save({"elo": elo, "ehi": ehi, "logits": [l.cpu() for l in out["classified_edges"]], "h": h.cpu()},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["g4_s02_L3", "g5_mean", "g5_max", "g8_train_topology_L3"])
def test_sharded_hip_forward_two_ranks_one_gpu(name, tmp_path):
    """The edge-partitioned path with the real HIP backend: 2 processes share this box's GPU, exchange through
    gloo, and together must reproduce the single-rank reference results."""
    import os
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_shard_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    c = Case(name)
    parts = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2)]
    for i in range(c.meta["n_out"]):
        full = torch.cat([p["logits"][i] for p in parts], 0)
        assert full.shape[0] == c.meta["E"]
        assert (full[c.sub_idx] - c.logits(i)).abs().max().item() <= LOGIT_TOL
    scale = max(1.0, c.h(f64=True).abs().max().item())
    for p in parts:
        assert (p["h"].double() - c.h(f64=True)).abs().max().item() <= 1e-4 * scale


def test_sharded_path_world_one_equals_monolithic():
    """Phase-by-phase path (un-fused h0, workspace region views) == the one-call forward."""
    import os
    import torch.distributed as dist
    from mtmc_mpn import distributed as mdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        c = Case("g4_s02_L3")
        m, g = c.model().cuda().eval(), to_gpu(c.graph())
        n, e = g.x.shape[0], g.edge_index.shape[1]
        with torch.no_grad():
            ref, href = m(g)
            out, h = mdist.sharded_forward(m, g.x, (0, n, n), g.edge_index, g.edge_attr, e)
        assert (out["classified_edges"][0] - ref["classified_edges"][0]).abs().max().item() <= 2e-6
        assert (h - href).abs().max().item() <= 1e-6 * href.abs().max().item()
    finally:
        dist.destroy_process_group()


def test_deterministic_mode_is_bitwise_repeatable_and_correct():
    """MTMC_F_DETERMINISTIC: per-chunk partial sums added in a fixed order instead of float atomics."""
    from oracle import mpn_oracle
    for name in ("g4_s02_L3", "g5_mean", "g8_train_topology_L3", "g2_random_L3_C3"):   # sorted, mean, block-sorted, unsorted
        c = Case(name)
        m, d = c.model(), c.graph()
        m = m.cuda().eval()
        m.deterministic = True
        g = to_gpu(d)
        with torch.no_grad():
            o1, h1 = m(g)
            o2, h2 = m(g)
        for i in range(c.meta["n_out"]):
            assert (o1["classified_edges"][i].cpu()[c.sub_idx] - c.logits(i)).abs().max().item() <= LOGIT_TOL, name
        scale = max(1.0, c.h(f64=True).abs().max().item())
        assert (h1.cpu().double() - c.h(f64=True)).abs().max().item() <= 1e-4 * scale
        if name == "g4_s02_L3":                # globally row-sorted: the fixed-order path is taken
            assert torch.equal(h1, h2)


def test_forked_edge_encoder_flag():
    """MTMC_F_FORK: the edge encoder's statistics pass on a side stream beside the node-encoder GEMMs; same results."""
    import ctypes as C
    from mtmc_mpn import _lib, engine
    c = Case("g4_s02_L3")
    m, g = c.model().cuda().eval(), to_gpu(c.graph())
    eng = engine.ForwardEngine(m)
    with torch.no_grad():
        prep = eng.prepare(g.x, g.edge_index, g.edge_attr)
        eng.set_flags(prep, _lib.F_FORK)
        for _ in range(2):
            _lib.check(eng.lib.mtmc_mpn_forward(C.byref(prep.model), C.byref(prep.call)))
        torch.cuda.synchronize()
        logits, h = eng.outputs(prep)
    assert (logits[0].cpu()[c.sub_idx] - c.logits(0)).abs().max().item() <= LOGIT_TOL
    assert (h.cpu().double() - c.h(f64=True)).abs().max().item() <= 1e-4 * max(1.0, c.h(f64=True).abs().max().item())


def test_captured_forward_replays():
    """model.capture(data): a HIP-graph replay gives the eager results, follows in-place input and weight changes."""
    c = Case("g4_s02_L3")
    m, g = c.model().cuda().eval(), to_gpu(c.graph())
    with torch.no_grad():
        eager, eager_h = m(g)
        e0 = eager["classified_edges"][0].clone()
        replay = m.capture(g)
        out, h = replay()
        assert (out["classified_edges"][0] - e0).abs().max().item() <= 2e-6
        g.edge_attr.mul_(1.5)                              # new input values, same buffers
        want, _ = m(g)
        w0 = want["classified_edges"][0].clone()
        out2, _ = replay()
        assert (out2["classified_edges"][0] - w0).abs().max().item() <= 2e-6
        assert (w0 - e0).abs().max().item() > 1e-4         # and it did change
    m.train()
    with pytest.raises(RuntimeError):
        m.capture(g)
