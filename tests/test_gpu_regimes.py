"""Dispatch regimes the kernel selection creates, each against the oracle (SURVEY.md 8(d) config 2b and friends).

The tracker-scale S02 graph -- cams 250/221/281/250 (the reference's misc/mtsc_BUPT21 counts for c007/c008, 250 for the
two cameras whose files are not shipped) in ONE graph, as `bs_test = 2000` makes it (main.py:86, inference.py:407-413):
N = 1002, E = 751 202 -- is the largest list the FEW-edge forms of the edge passes take since round 5 (<= 1 572 864 edges: one edge
per thread in grid-stride loops, e' stored, ONE any-order matrix-core pass C); a second graph of the same kind, four cameras of
370 tracklets (N = 1480, E = 1 642 800), is where FEW node rows (the few-row encoder kernels) meet the MANY-edge forms (four edges
per thread in passes A/B, e' never stored, the sorted matrix-core pass C with the walk launched behind it).
On top of both, the same graphs
  * with a seeded edge permutation: rows unsorted -> above the threshold the sorted matrix-core kernel returns at once on the
    device-side flag and the half-wave walk does every round; below it the any-order kernel takes masked passes / per-edge atomics;
  * with nodes interleaved over the cameras, i.e. the TRAINING order (train.py:295-302, :323-329): rows sorted inside
    each camera block only;
  * in deterministic mode (fixed-order aggregation; bitwise repeatable h).
Tolerance: 1e-4 on the logits (north_star), labels equal outside the 2e-4 margin guard, h within 1e-4 relative.
"""
import copy
import types

import pytest
import torch

import mtmc_mpn
from golden_util import ARCH
from mtmc_mpn import _lib, engine, graphs

pytestmark = pytest.mark.gpu
LOGIT_TOL, MARGIN_GUARD = 1e-4, 2e-4


def _build(cam_of_node=None, seed=2):
    """camera_graph with the 16 KB-per-edge attribute gathers done on the device (inputs are inputs: the oracle gets the
    same tensors)."""
    import torch.nn.functional as F
    if cam_of_node is None:
        cam_of_node = torch.repeat_interleave(torch.arange(4), torch.tensor(list(graphs.S02_TRACKER_CAMS)))
    n = cam_of_node.numel()
    x = F.normalize(torch.randn(n, 2048, generator=torch.Generator().manual_seed(seed)), p=2, dim=0)
    ei = graphs.camera_edge_index(cam_of_node).contiguous()
    ea = graphs.appearance_edge_attr(x.cuda(), ei.cuda()).cpu()
    return types.SimpleNamespace(x=x, edge_index=ei, edge_attr=ea)


SMALL_EDGES = 6144 * 256          # csrc/kernels.h kSmallEdges: the few-edge forms of the edge passes up to here


@pytest.fixture(scope="module")
def tracker():
    d = _build()
    assert d.x.shape[0] == 1002 and d.edge_index.shape[1] == 751_202
    assert bool((d.edge_index[0][1:] >= d.edge_index[0][:-1]).all())       # inference order: rows non-decreasing
    return d


@pytest.fixture(scope="module")
def many_edge():
    d = _build(cam_of_node=torch.repeat_interleave(torch.arange(4), torch.tensor([370] * 4)), seed=3)
    assert d.x.shape[0] == 1480 and d.edge_index.shape[1] == 1_642_800 > SMALL_EDGES
    return d


@pytest.fixture(params=["tracker", "many_edge"])
def big(request):
    return request.getfixturevalue(request.param)


def _model(L, Cs=1, **over):
    torch.manual_seed(0)
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=Cs, **over)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, sd, params


def _oracle(sd, params, d, dtype=torch.float64):
    from oracle import mpn_oracle
    with torch.no_grad():
        out, h = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, d.x, d.edge_index, d.edge_attr, dtype=dtype)
    return out["classified_edges"], h


def _gpu(m, d, transposed_view=True):
    ei = d.edge_index.t().contiguous().cuda().t() if transposed_view else d.edge_index.cuda()
    with torch.no_grad():
        out, h = m(types.SimpleNamespace(x=d.x.cuda(), edge_index=ei, edge_attr=d.edge_attr.cuda()))
    torch.cuda.synchronize()
    return [t.cpu() for t in out["classified_edges"]], h.cpu()


def _check(got, h, want64, h64, tag):
    for i, (g, w) in enumerate(zip(got, want64)):
        err = (g.double() - w).abs().max().item()
        assert err <= LOGIT_TOL, f"{tag}[{i}]: max |dlogit| {err:.3e}"
        margin = w[:, 1] - w[:, 0]
        guard = margin.abs() > MARGIN_GUARD
        flips = int((((g[:, 1] - g[:, 0]) > 0) != (margin > 0))[guard].sum())
        assert flips == 0, f"{tag}[{i}]: {flips} label flips outside the margin guard"
    scale = max(1.0, h64.abs().max().item())
    assert (h.double() - h64).abs().max().item() <= 1e-4 * scale, tag


@pytest.mark.parametrize("L,Cs", [(1, 1), (3, 1), (3, 3)])
def test_tracker_scale_s02_against_fp64_oracle(tracker, L, Cs):
    """SURVEY 8(d) config 2b at the shipped L = 1 and at the metric's L = 3 (and every step classified)."""
    m, sd, params = _model(L, Cs)
    plan = engine.ForwardEngine(m).plan(1002, 751_202)
    assert plan.pass_c == _lib.PASS_C_MFMA_ANY and not plan.lazy_edges and plan.edges_per_thread == 1     # few-edge forms, in
    assert plan.enc_kernel[0] == _lib.GEMM_FEW_L0 and plan.enc_kernel[1] == _lib.GEMM_FEW_WAVE            # grid-stride loops
    got, h = _gpu(m.cuda(), tracker)
    want64, h64 = _oracle(sd, params, tracker)
    assert len(got) == Cs
    _check(got, h, want64, h64, f"tracker L={L} Cs={Cs}")
    # ... and no further from fp64 than the fp32 reference path itself (x4 slack), as for the fixtures
    want32, _ = _oracle(sd, params, tracker, torch.float32)
    ref_err = (want32[-1].double() - want64[-1]).abs().max().item()
    err = (got[-1].double() - want64[-1]).abs().max().item()
    assert err <= max(4 * ref_err, 2e-5), f"|gpu - fp64| {err:.2e} vs the fp32 reference's own {ref_err:.2e}"


def test_many_edge_forms_meet_the_few_row_encoder(many_edge):
    """1480 rows / 1 642 800 edges at the metric's L = 3: the few-row encoder kernels with the many-edge edge passes."""
    m, sd, params = _model(3, 1)
    plan = engine.ForwardEngine(m).plan(1480, 1_642_800)
    assert plan.pass_c == _lib.PASS_C_MFMA_SORTED and plan.lazy_edges and plan.edges_per_thread == 4
    assert plan.enc_kernel[0] == _lib.GEMM_FEW_L0 and plan.enc_kernel[1] == _lib.GEMM_FEW_WAVE
    got, h = _gpu(m.cuda(), many_edge)
    want64, h64 = _oracle(sd, params, many_edge)
    _check(got, h, want64, h64, "1480 / 1642800 L=3")


@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_unsorted_rows_of_long_lists(big, agg):
    """A seeded permutation of the edges.  Above the regime threshold prep_kernel flags the rows as unsorted, the sorted
    matrix-core kernel returns at once and pass_c_kernel (the walk, atomics per short run) does every round; below it the
    any-order matrix-core kernel takes the list as it comes.  Logits must match the sorted run's after un-permuting (G7's
    property, at these sizes) and the oracle on the permuted list."""
    tracker = big
    m, sd, params = _model(3, 1, node_agg_fn=agg)
    m = m.cuda()
    sorted_logits, sorted_h = _gpu(m, tracker)
    perm = torch.randperm(tracker.edge_index.shape[1], generator=torch.Generator().manual_seed(11))
    d = types.SimpleNamespace(x=tracker.x, edge_index=tracker.edge_index[:, perm].contiguous(), edge_attr=tracker.edge_attr[perm])
    assert not bool((d.edge_index[0][1:] >= d.edge_index[0][:-1]).all())
    got, h = _gpu(m, d, transposed_view=False)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, f"permuted {agg}")
    assert (got[0] - sorted_logits[0][perm]).abs().max().item() <= 2e-5       # same graph, same answer per edge
    assert (h - sorted_h).abs().max().item() <= 1e-4 * max(1.0, sorted_h.abs().max().item())


def test_training_order_rows_block_sorted_at_tracker_scale():
    """Nodes ordered by identity, cameras interleaved (train.py:295-302): the edge list is a concatenation of per-camera
    blocks (train.py:323-329), rows ascending inside a block only -- unsorted for the kernels, with long sorted runs."""
    cam = torch.arange(1480) % 4
    d = _build(cam_of_node=cam, seed=7)
    row = d.edge_index[0]
    assert d.edge_index.shape[1] > SMALL_EDGES and not bool((row[1:] >= row[:-1]).all())
    m, sd, params = _model(3, 3)
    got, h = _gpu(m.cuda(), d)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, "training order")


def test_deterministic_mode_above_524288_edges(big):
    """MTMC_F_DETERMINISTIC on a long sorted list: per-span partials + agg_fixup_kernel instead of float atomics, on the
    matrix-core kernel (pass_c_sorted_kernel<., true>; 751 202 % 64 = 34: the partial last chunk is its own, too) -- with e'
    stored (751 202 edges) and recomputed (1 642 800).  h and the logits must be BITWISE equal between runs, and right."""
    tracker = big
    m, sd, params = _model(3, 1)
    m = m.cuda()
    m.deterministic = True
    assert engine.ForwardEngine(m).plan(tracker.x.shape[0], tracker.edge_index.shape[1], flags=_lib.F_DETERMINISTIC).pass_c == _lib.PASS_C_MFMA_SORTED
    got1, h1 = _gpu(m, tracker)
    got2, h2 = _gpu(m, tracker)
    assert torch.equal(h1, h2) and torch.equal(got1[0], got2[0])
    want64, h64 = _oracle(sd, params, tracker)
    _check(got1, h1, want64, h64, "deterministic")
    m.deterministic = False
    got3, h3 = _gpu(m, tracker)
    assert (h3 - h1).abs().max().item() <= 1e-5 * max(1.0, h1.abs().max().item())


def test_max_aggregation_and_reattach_at_tracker_scale(big):
    """The variants that stay on the walk / the MODE 2-3 pass A, at long-list sizes (fixtures cover them at <= 150k)."""
    tracker = big
    for over in (dict(node_agg_fn="max"), dict(reattach_initial_edges=True, reattach_initial_nodes=True)):
        m, sd, params = _model(2, 2, **over)
        got, h = _gpu(m.cuda(), tracker)
        want64, h64 = _oracle(sd, params, tracker)
        _check(got, h, want64, h64, str(over))


# ---- both sides of every dispatch threshold -----------------------------------------------------------------------------
# The library switches kernels at: 4096 node rows (pre-split layer 0, role-split layer 1, row-streaming layer 3 vs the in-loop
# kernel), 49 152 node rows (role-split layer 2 with 256-row tiles), 1 572 864
# edges (edges per thread, lazy e', which pass-C combination), 32 768 edges and 24 edges per source row (matrix-core pass C
# or the walk).  A graph just below and just above each of them must give the oracle's answer, and the plan query must
# say the two sides really take different kernels (otherwise the test tests nothing).

def _random_sorted_graph(n, pairs, seed):
    d = graphs.stress_graph(n, pairs, seed=seed)
    return types.SimpleNamespace(x=d.x, edge_index=d.edge_index, edge_attr=d.edge_attr)


THRESHOLDS = [
    # (name, graph below, graph above, plan attribute that must differ)
    ("1536 node rows", (1500, 30_000), (1600, 30_000), "enc_kernel"),
    ("4096 node rows", (4000, 60_000), (4200, 60_000), "enc_kernel"),
    ("49152 node rows", (49_000, 150_000), (49_300, 150_000), "enc_kernel"),
    ("1572864 edges", (3000, 780_000), (3000, 792_000), "edges_per_thread"),
    ("32768 edges", (500, 15_000), (500, 18_000), "pass_c"),
    ("24 edges per source row", (2000, 22_000), (2000, 26_000), "pass_c"),
]


@pytest.mark.parametrize("name,below,above,attr", THRESHOLDS, ids=[t[0] for t in THRESHOLDS])
def test_both_sides_of_a_dispatch_threshold(name, below, above, attr):
    m, sd, params = _model(2, 2)
    eng = engine.ForwardEngine(m)
    plans = []
    for i, (n, pairs) in enumerate((below, above)):
        d = _random_sorted_graph(n, pairs, seed=40 + i)
        plans.append(getattr(eng.plan(n, d.edge_index.shape[1]), attr))
        got, h = _gpu(m.cuda(), d, transposed_view=False)
        want64, h64 = _oracle(sd, params, d)
        _check(got, h, want64, h64, f"{name}: N={n} E={d.edge_index.shape[1]}")
        m.cpu()
    assert plans[0] != plans[1], f"{name}: both graphs take the same kernels ({attr} = {plans[0]})"


@pytest.mark.parametrize("n", [8192, 8300, 9001, 12345])
def test_many_row_encoder_with_ragged_tile_heights(n):
    """The role-split layer-1 kernel picks its tile height (80..128 rows) per launch and the pre-split layer-0 kernel runs
    256-row tiles: node counts that leave ragged last tiles, at a size the CPU oracle evaluates in seconds."""
    m, sd, params = _model(1, 1)
    plan = engine.ForwardEngine(m).plan(n, 2 * 40_000)
    assert plan.enc_kernel[:2] == [_lib.GEMM_PRESPLIT_256, _lib.GEMM_STAGED_128]
    d = _random_sorted_graph(n, 40_000, seed=n)
    got, h = _gpu(m.cuda(), d, transposed_view=False)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, f"N={n}")


def test_pipelined_layer0_panels_eager_and_captured():
    """Layer 0 of a many-row graph runs in row panels, the operand split of panel i + 1 on a side stream beside panel i's GEMM
    (api_internal.h l0_panels; round 4).  30 000 rows = two panels with a ragged second one (the full-size configs take 3 and
    10): against the fp64 oracle, and the same forward recorded into a HIP graph (the side stream's work joins the capture
    through the events) must replay to the same numbers, also after the inputs changed."""
    m, sd, params = _model(1, 1)
    n = 30_000
    d = _random_sorted_graph(n, 60_000, seed=5)
    m = m.cuda()
    got, h = _gpu(m, d, transposed_view=False)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, "pipelined layer 0, eager")
    data = types.SimpleNamespace(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda())
    with torch.no_grad():
        replay = m.capture(data)
        out, h2 = replay()
        torch.cuda.synchronize()
        assert (out["classified_edges"][0].cpu() - got[0]).abs().max().item() <= 2e-6
        data.x.mul_(1.5)                                  # new values in the captured input tensor
        want, want_h = m(types.SimpleNamespace(x=data.x.clone(), edge_index=data.edge_index, edge_attr=data.edge_attr))
        want = want["classified_edges"][0].clone()
        out, h2 = replay()
        torch.cuda.synchronize()
        assert (out["classified_edges"][0] - want).abs().max().item() <= 2e-6


@pytest.mark.parametrize("over", [dict(reattach_initial_nodes=True), dict(reattach_initial_nodes=True, reattach_initial_edges=True),
                                  dict(node_agg_fn="mean"), dict(node_agg_fn="max")],
                         ids=["reattach_nodes", "reattach_both", "mean", "max"])
def test_many_node_variants(over):
    """The matrix-core projection kernel (node rows >= 4096) in its 64-wide [h0 | h] form, with mean aggregation's 1/deg
    scaling on the way in, and beside the walk (max aggregation) -- the fixtures cover these variants at <= 450 nodes only."""
    m, sd, params = _model(3, 2, **over)
    d = _random_sorted_graph(4500, 70_000, seed=77)
    got, h = _gpu(m.cuda(), d, transposed_view=False)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, str(over))


# ---- the sorted-list matrix-core pass C (pass_c_sorted_kernel) on lists whose 32-edge groups are NOT one or two rows ------
def _mixed_degree_graph(n_dense, pairs_dense, n_sparse, pairs_sparse, seed, drop_tail=0):
    """Row-sorted list of > 524 288 (270 000 dense pairs) or > 1 572 864 edges (800 000): a dense block of nodes (hundreds of edges per row: one- and two-row groups) followed
    by a sparse block (a handful of edges per row: groups of 3-4 rows take the masked passes, groups of more rows the
    per-register atomics), and isolated nodes in between.  drop_tail trims the list so that E % 64 takes a chosen value."""
    g = torch.Generator().manual_seed(seed)
    n = n_dense + n_sparse

    def pairs(lo, cnt, m):
        a = torch.randint(0, cnt, (m,), generator=g)
        b = torch.randint(0, cnt - 1, (m,), generator=g)
        b = b + (b >= a).to(b.dtype)
        return a + lo, b + lo
    a1, b1 = pairs(0, n_dense, pairs_dense)
    a2, b2 = pairs(n_dense, n_sparse, pairs_sparse)
    a, b = torch.cat([a1, a2]), torch.cat([b1, b2])
    key, _ = torch.sort(torch.cat([a * n + b, b * n + a]))
    if drop_tail:
        key = key[:key.numel() - drop_tail]
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, 2048, generator=g)
    ea = torch.rand(ei.shape[1], 2, generator=g)
    return types.SimpleNamespace(x=x, edge_index=ei, edge_attr=ea)


@pytest.mark.parametrize("tail", [0, 37])
def test_deterministic_sorted_list_with_low_degree_stretches(tail):
    """The same kind of list in deterministic mode: rows that span several spans, rows strictly inside one, groups of many
    rows (all through masked passes there), a tail: bitwise repeatable, equal to the default mode within rounding, right."""
    d = _mixed_degree_graph(1500, 270_000, 6000, 9_000, seed=78)
    E = d.edge_index.shape[1]
    drop = (E - tail) % 64
    if drop:
        d = types.SimpleNamespace(x=d.x, edge_index=d.edge_index[:, :E - drop].contiguous(), edge_attr=d.edge_attr[:E - drop].contiguous())
    E = d.edge_index.shape[1]
    m, sd, params = _model(2, 2)
    m = m.cuda()
    m.deterministic = True
    assert engine.ForwardEngine(m).plan(d.x.shape[0], E, flags=_lib.F_DETERMINISTIC).pass_c == _lib.PASS_C_MFMA_SORTED
    got1, h1 = _gpu(m, d, transposed_view=False)
    got2, h2 = _gpu(m, d, transposed_view=False)
    assert torch.equal(h1, h2) and all(torch.equal(a, b) for a, b in zip(got1, got2))
    want64, h64 = _oracle(sd, params, d)
    _check(got1, h1, want64, h64, f"deterministic, mixed degrees, E={E}")
    m.deterministic = False
    got3, h3 = _gpu(m, d, transposed_view=False)
    assert (h3 - h1).abs().max().item() <= 1e-5 * max(1.0, h1.abs().max().item())


@pytest.mark.parametrize("tail", [0, 1, 37])
def test_sorted_many_edge_list_with_low_degree_stretches(tail):
    d = _mixed_degree_graph(1500, 800_000, 6000, 9_000, seed=77)
    E = d.edge_index.shape[1]
    drop = (E - tail) % 64                                   # leave exactly `tail` edges behind the last whole 64-edge chunk
    if drop:
        d = types.SimpleNamespace(x=d.x, edge_index=d.edge_index[:, :E - drop].contiguous(), edge_attr=d.edge_attr[:E - drop].contiguous())
    E = d.edge_index.shape[1]
    assert E > SMALL_EDGES and E % 64 == tail
    m, sd, params = _model(2, 2)
    plan = engine.ForwardEngine(m).plan(d.x.shape[0], E)
    assert plan.pass_c == _lib.PASS_C_MFMA_SORTED, plan.pass_c
    got, h = _gpu(m.cuda(), d, transposed_view=False)
    want64, h64 = _oracle(sd, params, d)
    _check(got, h, want64, h64, f"mixed degrees, E={E} (E % 64 = {tail})")
    deg = torch.bincount(d.edge_index[0], minlength=d.x.shape[0])
    assert int((deg == 0).sum()) > 0 and int(((deg > 0) & (deg < 8)).sum()) > 1000     # the stretches the test is about


def test_two_host_threads_run_many_row_forwards_on_their_own_streams():
    """the pipelined layer 0 had ONE side stream + event set per device; two host threads recorded and waited
    on the same event objects and a panel GEMM of one could start before its own operand split had finished.  Now one set
    per (device, caller stream): two threads, two streams, different inputs, interleaved -- each must get its own answer."""
    import threading
    m, sd, params = _model(1, 1)
    m = m.cuda()
    n = 40_000                                            # two row panels (1 + 2 rounds of workgroups)
    graphs_ = [_random_sorted_graph(n, 30_000, seed=70 + i) for i in range(2)]
    data = [types.SimpleNamespace(x=g.x.cuda(), edge_index=g.edge_index.cuda(), edge_attr=g.edge_attr.cuda()) for g in graphs_]
    assert engine.ForwardEngine(m).plan(n, data[0].edge_index.shape[1]).layer0_panels > 1
    with torch.no_grad():
        want = [m(d)[0]["classified_edges"][-1].clone() for d in data]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(12):
                    out = m(data[i])[0]["classified_edges"][-1]
                st.synchronize()
            got[i] = out
        except Exception as ex:   # noqa: BLE001
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i in range(2):
        assert (got[i] - want[i]).abs().max().item() <= 2e-5, f"thread {i}"
