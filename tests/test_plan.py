"""mtmc_mpn_plan_call (host-only): a SHARD of a big graph must take the kernels the whole graph takes.

Round 2's pass-C dispatch divided the LOCAL edge count by the GLOBAL node count, so the 8-way partition of BASELINE
config 5 (12.5 M edges per rank, 1 M nodes) silently fell back from the matrix-core pass C to the half-wave walk -- on
exactly the run north_star asks to scale.  Results are identical either way, so only a plan query can see it.  No GPU
needed: the query launches nothing and reads no pointer of the call."""
import copy

import pytest

import mtmc_mpn
from mtmc_mpn import _lib, distributed as mdist, engine


@pytest.fixture(scope="module")
def eng():
    params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
    return engine.ForwardEngine(mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval())


N5, E5 = 1_000_000, 100_000_000          # BASELINE config 5


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_config5_shards_take_the_whole_graphs_kernels(eng, world):
    whole = eng.plan(N5, E5)
    assert whole.pass_c == _lib.PASS_C_MFMA_SORTED and whole.lazy_edges and whole.edges_per_thread == 4
    assert whole.pass_a_col_blocks == 8      # 16 MB of column projections: pass A by 2 MB column blocks (one per XCD)
    assert whole.enc_kernel == [_lib.GEMM_PRESPLIT_256, _lib.GEMM_STAGED_128, _lib.GEMM_STAGED_128, _lib.GEMM_ROWS_16]
    rows = mdist.even_ranges(N5, world)
    edges = mdist.even_ranges(E5, world)
    for r in range(world):
        e_loc = edges[r][1] - edges[r][0]
        # row-complete shard (bench.py --gpus N: row-sorted list, boundaries snapped to row changes, own_rows)
        own = eng.plan(N5, e_loc, E5, node_range=rows[r], row_range=rows[r])
        assert own.avg_degree == pytest.approx(100.0, rel=1e-6)
        # general shard (node state all-reduced, every rank projects every node)
        gen = eng.plan(N5, e_loc, E5, node_range=rows[r])
        assert gen.avg_degree == pytest.approx(100.0)
        for p in (own, gen):
            assert p.pass_c == whole.pass_c, f"rank {r}/{world} left the matrix-core pass C"
            assert p.lazy_edges == whole.lazy_edges and p.edges_per_thread == whole.edges_per_thread
            assert p.pass_a_col_blocks == whole.pass_a_col_blocks, f"rank {r}/{world} left the column-blocked pass A"
            assert p.enc_kernel == whole.enc_kernel and p.enc_split_k == whole.enc_split_k


def test_row_complete_shard_is_judged_on_its_own_rows(eng):
    # a rank whose rows have few out-edges must NOT be sent to the matrix-core kernel just because the graph's mean is high
    p = eng.plan(100_000, 600_000, 10_000_000, node_range=(0, 50_000), row_range=(0, 50_000))
    assert p.avg_degree == pytest.approx(12.0) and p.pass_c == _lib.PASS_C_WALK
    # ... and an empty row range plans without dividing by zero
    p = eng.plan(100_000, 0, 10_000_000, node_range=(0, 0), row_range=(7, 7))
    assert p.avg_degree == 0.0 and p.pass_c == _lib.PASS_C_WALK


def test_single_gpu_regimes(eng):
    s02 = eng.plan(450, 150_454)                       # headline graph: few rows, few edges
    # round 5: the few-row kernels (one launch per layer, K never cut across workgroups) where the call has a weight-plane cache
    assert s02.enc_kernel == [_lib.GEMM_FEW_L0] + [_lib.GEMM_FEW_WAVE] * 3 and s02.enc_split_k == [1, 1, 1, 1]
    assert s02.pass_c == _lib.PASS_C_MFMA_ANY and not s02.lazy_edges and s02.edges_per_thread == 1
    old = eng.plan(450, 150_454, weight_cache=False)   # ... without one (plain C callers, cache_weight_planes = False): rounds 1-4
    assert old.enc_kernel == [_lib.GEMM_INLOOP_64] * 4 and old.enc_split_k[:3] == [4, 4, 4] and old.enc_split_k[3] == 1
    assert old.enc2_passenger and s02.enc2_passenger
    trk = eng.plan(1002, 751_202)                      # since round 5 on the few-edge forms (<= 1572864 edges)
    big = eng.plan(1480, 1_642_800)                    # four cameras of 370: few rows AND the many-edge forms
    assert trk.enc_kernel == [_lib.GEMM_FEW_L0] + [_lib.GEMM_FEW_WAVE] * 3
    trk_old = eng.plan(1002, 751_202, weight_cache=False)
    assert trk_old.enc_kernel == [_lib.GEMM_INLOOP_64] * 4 and trk_old.enc_split_k == [1, 4, 4, 1]   # 256 tiles in layer 0: unsplit
    mid = eng.plan(2000, 100_000)                      # more rows than the few-row kernels take, fewer than the many-row ones
    assert mid.enc_kernel == [_lib.GEMM_INLOOP_64] * 4
    shard = eng.plan(450, 75_000, 150_454, node_range=(0, 225))    # a rank of a 2-way split of the headline graph
    assert shard.enc_kernel == s02.enc_kernel
    assert trk.pass_c == _lib.PASS_C_MFMA_ANY and not trk.lazy_edges and trk.edges_per_thread == 1
    assert big.enc_kernel == trk.enc_kernel
    assert big.pass_c == _lib.PASS_C_MFMA_SORTED and big.lazy_edges and big.edges_per_thread == 4
    cfg4 = eng.plan(100_000, 10_000_000)
    assert cfg4.enc_kernel[0] == _lib.GEMM_PRESPLIT_256 and cfg4.pass_c == _lib.PASS_C_MFMA_SORTED
    assert cfg4.pass_a_col_blocks == 0 and s02.pass_a_col_blocks == 0      # 1.6 MB of Pc fit an XCD's L2: edge order
    # round 4: the headline graph's 21 launches -- enc2 rides in the last encoder layer's launch, no node_stat launches -- and
    # the many-row graphs' pipelined layer 0 (3 panels of 1, 2, 4 rounds at config 4; 10 at config 5)
    assert s02.enc2_passenger and s02.node_stat_folded and s02.layer0_panels == 1
    assert not cfg4.enc2_passenger and not cfg4.node_stat_folded and cfg4.layer0_panels == 3
    assert eng.plan(1_000_000, 100_000_000).layer0_panels == 10 and not big.node_stat_folded and big.enc2_passenger is False
    assert trk.node_stat_folded and trk.enc2_passenger
    assert eng.plan(1_000_000, 20_000_000).pass_a_col_blocks == 0          # 20 edges per row: sub-runs too short for 8 blocks
    assert eng.plan(1_000_000, 100_000_000, training=True).pass_a_col_blocks == 0
    det = eng.plan(100_000, 10_000_000, flags=_lib.F_DETERMINISTIC)
    assert det.pass_c == _lib.PASS_C_MFMA_SORTED        # many edges: the sorted kernel's fixed-order variant (no atomics)
    det_small = eng.plan(450, 150_454, flags=_lib.F_DETERMINISTIC)
    assert det_small.pass_c == _lib.PASS_C_WALK         # few edges: fixed-order aggregation lives in the walk
    trn = eng.plan(440, 180_000, training=True)
    assert trn.pass_c == _lib.PASS_C_MFMA_ANY and not trn.lazy_edges  # Dropout compiled into the few-edge matrix-core pass C; e' kept for the tape
    assert eng.plan(1480, 1_642_800, training=True).pass_c == _lib.PASS_C_WALK       # (many-edge lists: Dropout stays on the walk)
    assert trn.enc_kernel == [_lib.GEMM_FEW_L0] + [_lib.GEMM_FEW_WAVE] * 3     # training forwards take the few-row kernels too (Dropout compiled in)
    assert eng.plan(440, 180_000, training=True, weight_cache=False).enc_kernel == [_lib.GEMM_INLOOP_64] * 4


def test_deterministic_mode_beyond_the_sorted_kernels_node_limit(eng):
    """Round-3 regression (ADVICE): with >= 2^24 GLOBAL node rows (reachable with row-sharded multi-GPU calls) a many-edge
    call runs the any-order matrix-core kernel, which only knows float atomics; a deterministic call must keep the walk
    there (its carry[] is what agg_fixup_kernel adds up), and the plan must name the kernel that really runs."""
    big_n = 1 << 24
    rows, e_loc, e_tot = (0, 1 << 21), 300_000_000, 2_000_000_000
    det = eng.plan(big_n, e_loc, e_tot, node_range=rows, row_range=rows, flags=_lib.F_DETERMINISTIC)
    assert det.avg_degree > 24 and det.pass_c == _lib.PASS_C_WALK
    plain = eng.plan(big_n, e_loc, e_tot, node_range=rows, row_range=rows)
    assert plain.pass_c == _lib.PASS_C_MFMA_ANY         # not MFMA_SORTED: launch_pass_c cannot take the sorted kernel here
    below = eng.plan(big_n - 1, e_loc, e_tot, node_range=rows, row_range=rows, flags=_lib.F_DETERMINISTIC)
    assert below.pass_c == _lib.PASS_C_MFMA_SORTED


def test_plan_rejects_inconsistent_ranges(eng):
    with pytest.raises(RuntimeError):
        eng.plan(100, 50, 10)                           # more local edges than the graph has
    with pytest.raises(RuntimeError):
        eng.plan(100, 50, node_range=(10, 200))
