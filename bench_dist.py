"""Multi-GPU leg of bench.py: the 1M-node / 100M-edge graph (BASELINE config 5) edge-range partitioned over
the ranks of one node, one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
Strong scaling: the graph is fixed and row-sorted, so every rank gets a row-complete edge shard (boundaries snapped to
row changes), encodes and projects exactly the node rows its edges start from, and keeps their node state to itself: per
round only the column projections Pc [N,4] are all-gathered (16 B per node instead of the 128 B of the node state), the
BatchNorm statistics are all-reduced at their dependency points (14 collectives per L = 3 forward, mtmc_mpn/distributed.py
step_plan), and the [N,32] node state is all-gathered once at the end.  value = total edges / step time, step time = max
over ranks between two barriers.  The line also carries `prediction`: the per-rank kernel time by phase scaled from the
committed 1-GPU profile and the expected time of every collective over xGMI, next to what was
measured.
"""
import copy
import json
import os
import time

import torch
import torch.distributed as dist

import mtmc_mpn
from mtmc_mpn import _lib, distributed as mdist, graphs

ARCH = "resnet101"


# ---- what the first real multi-GPU run is held against -------------------------------------------
XGMI_LINK_GBS = 153.0          # one xGMI link, one direction (7 per GPU, one to every peer); the run is priced at
XGMI_EFF = 0.65                # 65 % of it for large messages -- a ring step moves one piece over one link
SMALL_COLLECTIVE_US = 20.0     # an RCCL all-reduce of a few KB (latency-bound; 15-25 us typical on one node)
LAUNCH_FLOOR_US = 5.0          # a dependent kernel at its floor (the S02 forward: 21 launches at 5-11 us)


def predict_scaling(world, n=1_000_000, e=100_000_000, L=3, stats_csv=None):
    """Expected per-rank time of one config-5 forward on `world` ranks (row-complete shards, own rows): kernel time by phase
    = the committed 1-GPU rocprofv3 averages (profiles/rNN_cfg5_kernel_stats.csv) / world -- every kernel of the path works on
    the rank's own edges or own rows; what does NOT shrink is named -- plus every collective priced over xGMI."""
    import csv
    import glob
    if stats_csv is None:
        # the PHASE path's kernels (tools/phase_loop.py: one operand-split pass + one layer-0 GEMM launch, what a rank runs), not
        # the one-call forward's (layer 0 in row panels beside the split on a side stream: round 4's table added those up
        # serially and missed the measured world-1 step by 8 %)
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        found = sorted(glob.glob(os.path.join(here, "r*_cfg5_phase_kernel_stats.csv"))) or \
            sorted(glob.glob(os.path.join(here, "r*_cfg5_kernel_stats.csv")))
        if not found:
            return None
        stats_csv = found[-1]
    avg, calls = {}, {}
    for r in csv.DictReader(open(stats_csv)):
        if r.get("AverageNs"):
            avg[r["Name"]] = float(r["AverageNs"]) * 1e-6          # ms per launch
            calls[r["Name"]] = int(r["Calls"])
    steps = calls.get("mtmc::prep_kernel<false>", calls.get("mtmc::prep_kernel", min(calls.values()) if calls else 1))   # one per forward

    def per_forward(prefix):                                          # all launches of the kernels named prefix*, per forward
        return sum(avg[k] * calls[k] / steps for k in avg if k.startswith(prefix))
    phases = {
        "encoder layer 0 (operand split + gemm_f16p_m16)": per_forward("mtmc::gemm_f16p") + per_forward("mtmc::split_rows"),
        "encoder layers 1-3 (gemm_staged, gemm_rows)": per_forward("mtmc::gemm_staged") + per_forward("mtmc::gemm_rows"),
        "prep + enc2 (edge branch) + weight-plane verification": per_forward("mtmc::prep") + per_forward("mtmc::enc2") + per_forward("mtmc::split_jobs") + per_forward("mtmc::colblock"),
        "pass_a x L (gathers Pc of ALL nodes: per-edge cost does not improve with P)": per_forward("mtmc::pass_a"),
        "pass_b x L": per_forward("mtmc::pass_b"),
        "pass_c x L": per_forward("mtmc::pass_c_sorted") + per_forward("mtmc::pass_c_kernel"),
        "node_proj + node_stat x L, h0 (own rows)": per_forward("mtmc::node_proj") + per_forward("mtmc::node_stat") + per_forward("mtmc::bn_relu_rows"),
    }
    one_gpu = sum(phases.values())
    n_launch = 14 + 5 * L + 1
    kernels = {k: v / world for k, v in phases.items()}
    kernel_ms = sum(kernels.values()) + (n_launch * LAUNCH_FLOOR_US * 1e-3 if world > 1 else 0.0) * (1 - 1 / world)
    bw = XGMI_LINK_GBS * XGMI_EFF * 1e9
    ring = lambda total_bytes: (world - 1) * (total_bytes / world) / bw * 1e3 + SMALL_COLLECTIVE_US * 1e-3   # ms
    coll = []
    if world > 1:
        coll.append({"what": "BatchNorm statistics all-reduce (fp64, 0.5-18 KB)", "count": 4 + 2 * L,
                     "bytes_each": "<= 18 KB", "ms_each": SMALL_COLLECTIVE_US * 1e-3})
        coll.append({"what": "all-gather of the column projections Pc [N,4] f32", "count": L, "bytes_each": 16 * n,
                     "ms_each": ring(16 * n)})
        coll.append({"what": "final all-gather of the node state [N,32] f32 (replicate_h=False drops it)", "count": 1,
                     "bytes_each": 128 * n, "ms_each": ring(128 * n)})
    coll_ms = sum(c["count"] * c["ms_each"] for c in coll)
    total = kernel_ms + coll_ms
    final_gather = coll[-1]["ms_each"] if coll else 0.0
    return {"world": world, "source": os.path.basename(stats_csv), "one_gpu_kernel_ms": round(one_gpu, 3),
            "predicted_ms_per_step_sharded_node_state": round(total - final_gather, 3),
            "per_rank_kernel_ms_by_phase": {k: round(v, 3) for k, v in kernels.items()},
            "per_rank_kernel_ms": round(kernel_ms, 3), "collectives": coll, "collectives_ms": round(coll_ms, 3),
            "n_collectives": sum(c["count"] for c in coll), "predicted_ms_per_step": round(total, 3),
            "predicted_edges_per_s": e / (total * 1e-3), "predicted_speedup_vs_1gpu": round(one_gpu / total, 2),
            "assumptions": f"xGMI link {XGMI_LINK_GBS} GB/s x {XGMI_EFF} efficiency, ring all-gather ((P-1)/P of the bytes over one "
                           f"link), {SMALL_COLLECTIVE_US} us per small all-reduce, nothing overlapped"}


def main_distributed(args):
    import bench
    import sys
    # stdout must carry ONE JSON line: RCCL prints a version banner to the C-level stdout at communicator creation, so
    # everything but the final line is sent to stderr (file descriptor 1 is pointed at 2 for the duration of the run)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    name = "cfg5" if args.workload == "auto" else args.workload
    desc, L, cs = bench.WORKLOADS[name]
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
    torch.manual_seed(0)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).to(device).eval()

    # every rank generates the same seeded graph on its own GPU, then keeps only its shard
    full = bench.make_workload(name, device)
    n, e = full.x.shape[0], full.edge_index.shape[1]
    elo, ehi = mdist.edge_ranges(full.edge_index[0], e, world, snap_to_rows=True)[rank]
    ei_loc = full.edge_index[:, elo:ehi].clone()
    ea_loc = full.edge_attr[elo:ehi].clone()
    rr = mdist.row_ranges_of(ei_loc)          # row-sorted + snapped: complete rows per rank -> only Pc [N,4] travels per round
    own = rr is not None                      # ... and the rank encodes exactly the rows it projects: h0 stays home too
    lo, hi = mdist.tile_rows(rr, n)[rank] if own else mdist.even_ranges(n, world)[rank]
    x_loc = full.x[lo:hi].clone()
    # parity of the sharded path against the ONE-GPU forward of the same graph (every rank holds the full graph at
    # this point): sampled edges of this rank's slice and sampled rows of the node state, checked after the timing
    with torch.no_grad():
        mono, mono_h = model(full)
        stride = max(1, (ehi - elo) // 200_000)
        ref_logits = mono["classified_edges"][-1][elo:ehi:stride].clone()
        ref_h = mono_h[::max(1, n // 50_000)].clone()
        del mono, mono_h
    model._engine = None
    from mtmc_mpn import torch_ops
    torch_ops._ENGINES.clear()                 # drops the monolithic call's workspace
    del full
    torch.cuda.empty_cache()

    def step():
        return mdist.sharded_forward(model, x_loc, (lo, hi, n), ei_loc, ea_loc, e, row_ranges=rr, own_rows=own)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(device)
        dist.barrier()
        t1 = time.perf_counter()
    t = torch.tensor([(t1 - t0) / args.steps], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sec = float(t.item())
    with torch.no_grad():
        out, h = step()
        d_logit = (out["classified_edges"][-1][::stride] - ref_logits).abs().max()
        d_h = (h[::max(1, n // 50_000)] - ref_h).abs().max() / ref_h.abs().max().clamp_min(1.0)
    par = torch.stack([d_logit.double(), d_h.double()])
    dist.all_reduce(par, op=dist.ReduceOp.MAX)
    parity = {"max_abs_logit_diff_vs_1gpu": float(par[0]), "max_rel_h_diff_vs_1gpu": float(par[1]),
              "edges_checked_per_rank": int(ref_logits.shape[0]), "tolerance": 1e-4}
    if not (parity["max_abs_logit_diff_vs_1gpu"] <= 1e-4 and parity["max_rel_h_diff_vs_1gpu"] <= 1e-4):
        raise RuntimeError(f"sharded forward differs from the 1-GPU forward: {parity}")

    # side measurement: the same forward with the node state left sharded (no final [N,32] all-gather) -- the callers of
    # the reference never read latent_node_feats; `value` above is the forward WITH the replicated output
    sec_sharded_h = None
    if rr is not None:
        with torch.no_grad():
            for _ in range(2):
                mdist.sharded_forward(model, x_loc, (lo, hi, n), ei_loc, ea_loc, e, row_ranges=rr, own_rows=own, replicate_h=False)
            torch.cuda.synchronize(device)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                mdist.sharded_forward(model, x_loc, (lo, hi, n), ei_loc, ea_loc, e, row_ranges=rr, own_rows=own, replicate_h=False)
            torch.cuda.synchronize(device)
            dist.barrier()
            t2 = torch.tensor([(time.perf_counter() - t0) / args.steps], dtype=torch.float64, device=device)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        sec_sharded_h = float(t2.item())

    # per-phase split (kernels vs collectives) with events, and what the HOST spends per phase (issuing the phase's
    # launches through mtmc_mpn_run_phase / issuing the collectives behind it), on every rank, a few extra iterations
    eng = model._engine
    with torch.no_grad():
        prep_evs = []
        host_ms = {}

        def host_add(key, t0):
            host_ms[key] = host_ms.get(key, 0.0) + (time.perf_counter() - t0) * 1e3

        SHORT = {_lib.PH_BEGIN: "begin", _lib.PH_EDGE_ENC: "edge_enc", _lib.PH_NODE_ENC: "enc", _lib.PH_NODE_COMBINE: "comb",
                 _lib.PH_NODE_H0: "h0", _lib.PH_ROUND_PROJ: "proj", _lib.PH_ROUND_A: "A", _lib.PH_ROUND_B: "B",
                 _lib.PH_ROUND_STAT: "stat", _lib.PH_ROUND_C: "C", _lib.PH_END: "end"}
        WITH_ARG = (_lib.PH_NODE_ENC, _lib.PH_NODE_COMBINE)

        class TimedBackend:
            """The engine's phase interface with a host clock around run_phase (everything else passes through)."""
            def __init__(self, inner):
                self._inner = inner

            def __getattr__(self, name):
                return getattr(self._inner, name)

            def run_phase_list(self, prep, pairs):
                t0 = time.perf_counter()
                self._inner.run_phase_list(prep, pairs)        # one library call: everything between two collectives
                host_add("+".join(SHORT.get(ph, str(ph)) + (str(arg) if ph in WITH_ARG else "") for ph, arg in pairs), t0)

        class Timed(mdist.ShardedForward):
            def _timed(self, fn, *a_):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                a.record(); fn(*a_); b.record()
                host_add("collectives", t0)
                prep_evs.append((a, b))

            def _sum(self, tns):
                self._timed(super()._sum, tns)

            def _max(self, tns):
                self._timed(super()._max, tns)

            def _gather_rows(self, *a_):
                self._timed(super()._gather_rows, *a_)
        timed = Timed(TimedBackend(eng), model.spec)
        reps = 3
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_host = time.perf_counter()
        s_ev.record()
        for _ in range(reps):
            timed(x_loc, (lo, hi, n), ei_loc, ea_loc, e, rr, own)
        e_ev.record()
        host_total = (time.perf_counter() - t_host) * 1e3 / reps      # host time to ISSUE one forward (no sync inside)
        torch.cuda.synchronize(device)
        coll = sum(a.elapsed_time(b) for a, b in prep_evs) / reps
        total = s_ev.elapsed_time(e_ev) / reps
        split = {"collectives_ms": coll, "kernels_and_gather_ms": total - coll, "step_ms_this_rank": total}
        mine = {"rank": rank, "host_issue_ms_per_forward": round(host_total, 4), "gpu_step_ms": round(total, 4),
                "gpu_collectives_ms": round(coll, 4),
                "host_ms_by_phase": {k: round(v / reps, 4) for k, v in sorted(host_ms.items())}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # the headline graph (AIC19-S02, N=450) on the same ranks: what BASELINE.json's metric names, although a
    # 150k-edge forward is latency-bound and gains nothing from more GPUs (reported beside the scaling workload)
    s02 = None
    if name != "s02":
        full = bench.make_workload("s02", device)
        n2, e2 = full.x.shape[0], full.edge_index.shape[1]
        elo2, ehi2 = mdist.edge_ranges(full.edge_index[0], e2, world, snap_to_rows=True)[rank]
        ei2, ea2 = full.edge_index[:, elo2:ehi2].clone(), full.edge_attr[elo2:ehi2].clone()
        rr2 = mdist.row_ranges_of(ei2)
        own2 = rr2 is not None
        lo2, hi2 = mdist.tile_rows(rr2, n2)[rank] if own2 else mdist.even_ranges(n2, world)[rank]
        x2 = full.x[lo2:hi2].clone()
        params2 = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=1)
        torch.manual_seed(0)
        model2 = mtmc_mpn.MOTMPNet(copy.deepcopy(params2), None, ARCH).to(device).eval()
        with torch.no_grad():
            for _ in range(5):
                mdist.sharded_forward(model2, x2, (lo2, hi2, n2), ei2, ea2, e2, row_ranges=rr2, own_rows=own2)
            torch.cuda.synchronize(device)
            dist.barrier()
            t0 = time.perf_counter()
            reps2 = 50
            for _ in range(reps2):
                mdist.sharded_forward(model2, x2, (lo2, hi2, n2), ei2, ea2, e2, row_ranges=rr2, own_rows=own2)
            torch.cuda.synchronize(device)
            dist.barrier()
            t2 = torch.tensor([(time.perf_counter() - t0) / reps2], dtype=torch.float64, device=device)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        s02 = {"workload": "s02 (N=450, E=150454, L=3) edge-partitioned over the same ranks", "ms_per_step": float(t2) * 1e3,
               "edges_per_s": e2 / float(t2),
               "note": "latency-bound graph: the 14 small collectives per forward outweigh the 0.17 ms of kernels"}

    # The headline is the forward whose node state stays with the rank that owns the rows (round 5): the reference's callers
    # discard latent_node_feats (inference.py:469 keeps only outputs['classified_edges']), and the final [N,32] all-gather is a
    # fifth of the predicted 8-rank step; the forward WITH the replicated output is reported beside it.
    sec_replicated = sec
    if sec_sharded_h is not None:
        sec = sec_sharded_h
    if rank == 0:
        b_fwd = bench.algorithmic_bytes_forward(n, e, L, cs)
        line = {"metric": "MPN forward edges/sec (+ achieved roofline fraction of the dominant kernel)",
                "value": e / sec, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic (seeded graph and features; random-init weights)",
                "config": {"workload": f"{name}: {desc}, L={L}, Cs={cs}, eval forward", "N": n, "E": e,
                           "parallelism": f"edge-range x{world} (rows of x range-partitioned for the encoder), "
                                          "RCCL all-reduce of BatchNorm statistics; " + ("row-complete shards: each rank keeps its rows' node state, per round only the column projections Pc [N,4] are all-gathered, the [N,32] node state once at the end" if rr is not None else "the [N,32] node state all-reduced per round")},
                "roofline": {"bound": "hbm", "achieved": b_fwd / sec / 1e9, "peak": bench.HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": b_fwd / sec / 1e9 / (bench.HBM_PEAK_GBS * world), "traffic": None,
                             "kernel": "whole forward (SURVEY 8(d) algorithmic bytes over the step time, all ranks)"},
                "cpu_baseline": None, "headline_graph_on_these_ranks": s02,
                "scaling_note": ("north_star asks for the S02 graph at 1 GPU and the 1M-node / 100M-edge graph at 1/2/4/8 GPUs: "
                                 "this line is the latter; its 1-GPU point is `scale_base.value` of the `--gpus 1` line "
                                 "(whose headline `value` is the S02 graph), so compare against that, not against the "
                                 "headline"),
                "forward_algorithmic": {"bytes": b_fwd, "GBps": b_fwd / sec / 1e9,
                                        "frac_of_aggregate_hbm_peak": b_fwd / sec / 1e9 / (bench.HBM_PEAK_GBS * world)},
                "rank0_split": split, "per_rank_host": per_rank, "parity_vs_1gpu": parity,
                "prediction": predict_scaling(world, n, e, L) if name == "cfg5" else None,
                "node_state": "sharded (each rank returns the rows it owns)" if sec_sharded_h is not None else "replicated",
                "ms_per_step_with_replicated_node_state": sec_replicated * 1e3}
        if line["prediction"]:
            pred = line["prediction"]["predicted_ms_per_step_sharded_node_state" if sec_sharded_h is not None else "predicted_ms_per_step"]
            line["prediction_over_measured"] = round(pred / (sec * 1e3), 4)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
