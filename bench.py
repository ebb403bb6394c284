#!/usr/bin/env python3
"""bench.py -- MPN-forward edges/s on MI355X, with the roofline of the dominant kernel and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|s02|s02_L1|s02_tracker|cfg4|cfg5]

A step = one full `MOTMPNet.forward` (all L rounds, eval mode, inputs resident in HBM).  Metric and
workloads are BASELINE.json's: at 1 GPU the AIC19-S02 graph (ground-truth topology 124/90/99/137
tracklets -> N=450, E=150 454; the real tracker files are not available offline) at L=3, Cs=1; the same
run also reports the 100k-node / 10M-edge stress graph (config 4) under "stress".  With --gpus N > 1 the
1M-node / 100M-edge graph (config 5) is edge-range partitioned over the ranks (strong scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import copy
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import _lib, graphs  # noqa: E402

ARCH = "resnet101"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # bf16 MFMA dense peak; the split kernel spends 6 bf16 products per fp32 product

PHASE_NAMES = {_lib.PH_BEGIN: "memset+prep_kernel(+split_rows_kernel)", _lib.PH_EDGE_ENC: "enc2_kernel", _lib.PH_NODE_ENC: "gemm_bn_kernel",
               _lib.PH_NODE_COMBINE: "combine_stats_kernel",
               _lib.PH_NODE_H0: "bn_relu_rows_kernel", _lib.PH_ROUND_PROJ: "node_proj_kernel",
               _lib.PH_ROUND_A: "pass_a_kernel", _lib.PH_ROUND_B: "pass_b_kernel", _lib.PH_ROUND_STAT: "node_stat_kernel",
               _lib.PH_ROUND_C: "pass_c_kernel", _lib.PH_END: "h_final_kernel"}

SMALL_EDGES = 6144 * 256        # csrc/kernels.h kSmallEdges: the few-edge forms of the edge passes up to here

WORKLOADS = {
    # name: (description, L, Cs)
    "s02": ("AIC19-S02 ground-truth topology, cams 124/90/99/137: N=450 E=150454", 3, 1),
    "s02_L1": ("AIC19-S02 ground-truth topology (shipped config L=1)", 1, 1),
    "s02_tracker": ("AIC19-S02 tracker-scale topology, cams 250/221/281/250: N=1002 E=751202", 3, 1),
    "cfg4": ("synthetic 100k nodes / 10M edges, sorted rows", 3, 1),
    "cfg5": ("synthetic 1M nodes / 100M edges, sorted rows", 3, 1),
}


def make_workload(name, device):
    if name in ("s02", "s02_L1"):
        d = graphs.camera_graph(graphs.S02_GT_CAMS, seed=2)
    elif name == "s02_tracker":
        d = graphs.camera_graph(graphs.S02_TRACKER_CAMS, seed=2)
    elif name == "cfg4":
        d = graphs.stress_graph(100_000, 5_000_000, seed=4, device=device)
    elif name == "cfg5":
        d = graphs.stress_graph(1_000_000, 50_000_000, seed=5, device=device)
    else:
        raise ValueError(name)
    ei = d.edge_index
    if ei.device.type == "cpu":
        ei = ei.t().contiguous().to(device).t() if not ei.is_contiguous() else ei.to(device)   # callers' [E,2].T view
    return types.SimpleNamespace(x=d.x.to(device), edge_index=ei, edge_attr=d.edge_attr.to(device))


def algorithmic_bytes_forward(n, e, L, cs, f=2048):
    """SURVEY.md 8(d), reference formulation (fp32 + int64): B_fwd = B_enc + L*B_round + Cs*B_cls."""
    return (f * 4 + 128) * n + 24 * e + L * (176 * e + 256 * n) + cs * 8 * e


def pmc_round(workload):
    """The newest round whose counter summaries for this workload are committed under profiles/."""
    for rnd in committed_rounds():
        if all(os.path.exists(os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc_{c}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE")):
            return rnd
    return None


def committed_rounds():
    """Round prefixes (rNN) that have files under profiles/, newest first."""
    import re
    names = os.listdir(os.path.join(ROOT, "profiles")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    return sorted({m.group(0) for m in (re.match(r"r\d\d", f) for f in names) if m}, reverse=True)


def pmc_traffic(workload, kernel, launches_per_step=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summaries (profiles/rNN_<workload>_pmc_*.txt,
    made by tools/pmc_summary.py from separate FETCH_SIZE and WRITE_SIZE passes of tools/fwd_loop.py on the same
    workload).  Units are KB; on gfx950 FETCH_SIZE counts wide (16 B/lane) streaming reads at one half
    (MI355X_MICROARCH.md, HBM), so reads are doubled; WRITE_SIZE is exact.  None when no summary is committed.
    `launches_per_step`: the kernel's launches per forward in the timing this is reported beside -- the profiled forward
    may cut the same work into more launches (layer 0 in row panels): all of a forward's launches are added up (forwards
    profiled = launches of prep_kernel) and divided by it, so that `traffic` and `achieved` are for the same unit of work."""
    tot = {}
    rnd = pmc_round(workload)
    if rnd is None:
        return None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{workload}_pmc_{ctr}.txt")
        num = den = forwards = 0.0
        for line in open(path):
            parts = line.split()
            if len(parts) >= 5 and ctr in parts:
                calls, avg = float(parts[-2]), float(parts[-1])
                if parts[0].startswith(kernel):
                    num += calls * avg
                    den += calls
                if parts[0].startswith("prep_kernel"):
                    forwards += calls
        if den == 0:
            return None
        tot[ctr] = num / (forwards * launches_per_step) if (launches_per_step and forwards) else num / den
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0


GEMM_KERNEL_NAMES = {_lib.GEMM_GENERIC: "linear_generic_kernel", _lib.GEMM_INLOOP_64: "gemm_bn_f16x3_kernel",
                     _lib.GEMM_INLOOP_128: "gemm_bn_f16x3_kernel", _lib.GEMM_PRESPLIT_256: "gemm_f16p_m16_kernel",
                     _lib.GEMM_STAGED_128: "gemm_staged_kernel", _lib.GEMM_ROWS_16: "gemm_rows_kernel",
                     _lib.GEMM_FEW_L0: "few_l0_kernel", _lib.GEMM_FEW_WAVE: "few_wave_kernel"}


def encoder_kernels(model, n, e):
    """Kernel name per node-encoder layer, asked of the library itself (mtmc_mpn_plan_call) instead of mirrored here."""
    from mtmc_mpn import engine
    eng = model._engine or engine.ForwardEngine(model)
    ids = eng.plan(n, e).enc_kernel
    names = [GEMM_KERNEL_NAMES[i] for i in ids]
    if os.environ.get("MTMC_GEMM_FP32"):
        names = ["gemm_bn_kernel" if i != _lib.GEMM_GENERIC else n_ for i, n_ in zip(ids, names)]
    elif os.environ.get("MTMC_GEMM_NO_F16"):
        names = [("gemm_bn_bf16x6_kernel" if i == _lib.GEMM_INLOOP_128 else "gemm_bn_kernel") if i != _lib.GEMM_GENERIC else n_
                 for i, n_ in zip(ids, names)]
    return names


def phase_cost(ph, arg, spec, n, e):
    """Algorithmic work of one launch of the phase's kernel: ('hbm', bytes) or ('mfma', flops).
    Per-kernel compulsory traffic of this build's formulation: streamed reads +
    writes, per-node tables counted once."""
    if ph == _lib.PH_NODE_ENC:
        lay = spec.enc_node[arg]
        return "mfma", 2.0 * n * lay.in_dim * lay.out_dim
    first = arg == 0
    e_in = 8 if (first and not spec.reattach_edges) else (24 if spec.reattach_edges and not first else 16 if not first else 8)
    table = {
        _lib.PH_NODE_COMBINE: 0.0,
        _lib.PH_BEGIN: 16 * e + 8 * e + 8 * e + 4 * n,                 # int64 row/col + attr in, int32 row/col out, degree
        _lib.PH_EDGE_ENC: 8 * e,
        _lib.PH_NODE_H0: 256 * n,
        _lib.PH_ROUND_PROJ: (128 + 32 + 128 + 128 + 4) * n,            # h in, P, Q out, cleared aggregation buffer, degree
        _lib.PH_ROUND_A: (8 + e_in + 16) * e + 32 * n,                 # row/col + e_in read, z1 written, P table once
        _lib.PH_ROUND_B: (4 + 16 + (0 if e > SMALL_EDGES else 16)) * e + (32 if e > SMALL_EDGES else 128) * n,   # row + z1 read; segment sums, or (few-edge graphs) e' written + one Q row per run
        _lib.PH_ROUND_STAT: (128 + 4 + 32) * n,
        _lib.PH_ROUND_C: (4 + 16) * e + (128 + 128) * n + (8 * e if arg >= spec.num_enc_steps - spec.num_class_steps else 0),
        _lib.PH_END: 256 * n,
    }
    return "hbm", float(table[ph])


def percentiles(v):
    v = sorted(v)
    n = len(v)
    return {"median": v[n // 2], "p10": v[int(0.1 * (n - 1))], "p90": v[int(round(0.9 * (n - 1)))], "min": v[0], "max": v[-1]}


def time_forward(model, data, steps, warmup, dist=None, fn=None):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides (the driver's contract):
    returns (seconds per step from the wall clock, per-step GPU milliseconds from HIP events recorded on the launch
    stream between consecutive steps -> median / p10 / p90)."""
    dev = data.x.device
    fn = fn or (lambda: model(data))
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            fn()
            marks[i + 1].record()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return (t1 - t0) / steps, percentiles(per_step)


def hbm_probe(device, gib=2, reps=20):
    """Measured HBM bandwidth of THIS box ("confirm on the box ... report against the measured peak too"):
    a device-to-device copy (read + write) and a triad a = b + s*c (two reads + one write) over buffers far larger
    than the 256 MB Infinity Cache; best of `reps`, HIP events."""
    n = gib * (1 << 30) // 4
    a, b, c = (torch.empty(n, dtype=torch.float32, device=device) for _ in range(3))
    b.fill_(1.0), c.fill_(2.0)
    out = {}
    for name, fn, nbytes in (("copy", lambda: a.copy_(b), 2 * 4 * n), ("triad", lambda: torch.add(b, c, alpha=0.5, out=a), 3 * 4 * n)):
        best = 1e30
        for _ in range(3):
            fn()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[name + "_GBps"] = nbytes / (best * 1e-3) / 1e9
    out["buffer_GiB"] = gib
    out["measured_peak_GBps"] = max(out["copy_GBps"], out["triad_GBps"])
    del a, b, c
    torch.cuda.empty_cache()
    return out


def time_phases(model, data, iters):
    """Per-kernel durations from HIP events recorded on the launch stream around every phase."""
    from mtmc_mpn import engine
    eng = model._engine or engine.ForwardEngine(model)
    seq = eng.phase_list()
    sums = [0.0] * len(seq)
    # (MTMC_PH_ROUND_STAT is a no-op on few-edge lists since round 4 -- node_stat_kernel's work moved into node_proj and
    # pass B there -- and is dropped from their report below: an empty phase would read as the 5 us event-pair floor)
    with torch.no_grad():
        prep = eng.prepare(data.x, data.edge_index, data.edge_attr)
        for it in range(iters + 2):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in seq]
            idx = [0]

            def before(ph, arg):
                evs[idx[0]][0].record()

            def after(ph, arg):
                evs[idx[0]][1].record()
                idx[0] += 1
            eng.run_phases(prep, after_phase=after, before_phase=before)
            torch.cuda.synchronize()
            if it >= 2:
                for i, (a, b) in enumerate(evs):
                    sums[i] += a.elapsed_time(b)
    folded = data.edge_index.shape[1] <= SMALL_EDGES
    keep = [i for i, (ph, _) in enumerate(seq) if not (folded and ph == _lib.PH_ROUND_STAT)]
    return [seq[i] for i in keep], [sums[i] / iters for i in keep]


def empty_event_pair_ms(n=200):
    """What a start/stop event pair reads with NOTHING between them on this box (about 5 us): the floor under every
    per-kernel figure of time_phases.  Reported beside the roofline, never subtracted (with a kernel in between part
    of it overlaps: rocprofv3's kernel duration sits between raw and raw - floor)."""
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in evs)
    return v[n // 2]


def cpu_baseline(name, params, sd, data, max_seconds=25.0):
    """The CPU oracle (bit-equal to the reference's CPU PyTorch path in the build container) timed on
    this host's cores, on a bounded sample of the workload."""
    from oracle import mpn_oracle
    # the GPU box hands one GPU's share of the host (16 cores) to this process; more threads than that
    # only oversubscribe ATen's small ops (measured: 256 threads are >100x slower than 16)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))
    e_full = data.edge_index.shape[1]
    if e_full <= 20_000_000:                 # the headline graph and config 4: the WHOLE workload ("configs 1-4 in full")
        x, ei, ea = data.x.cpu(), data.edge_index.cpu(), data.edge_attr.cpu()
        sample = "full workload"
    else:                                    # config 5 would need ~80 GB of host intermediates and minutes per forward
        scale = 100
        d = graphs.stress_graph(data.x.shape[0] // scale, e_full // (2 * scale), seed=4)
        x, ei, ea = d.x, d.edge_index, d.edge_attr
        sample = f"same recipe at 1/{scale} scale: {x.shape[0]} nodes / {ei.shape[1]} edges (edges/s is size-normalised)"
    # ~8 s per forward at config 4 on 16 threads: one warm-up and two timed forwards there, 3 + 10 on the headline graph
    n_warm, n_max = (1, 3) if e_full > 1_000_000 else (3, 13)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    times = []
    with torch.no_grad():
        t_start = time.perf_counter()
        for i in range(n_max):
            t0 = time.perf_counter()
            mpn_oracle.forward(sd_cpu, copy.deepcopy(params), ARCH, x, ei, ea)
            dt = time.perf_counter() - t0
            if i >= n_warm:
                times.append(dt)
            if time.perf_counter() - t_start > max_seconds and len(times) >= 2:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": ei.shape[1] / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample}; median of {len(times)} forwards after {n_warm} warm-up(s), {med * 1e3:.1f} ms each",
            "host_cpu_count": os.cpu_count()}


def panel_times(model, data, reps=5):
    """Layer-0 GEMM launches of the ONE-CALL forward (row panels beside the operand split on a side stream), timed by HIP
    events the library records around each of them on the launch stream (mtmc_dbg_panel_timing): per forward
    (sum of the panel launches in ms, number of panels), averaged over `reps` forwards.  None on graphs whose layer 0 is not
    the pre-split kernel."""
    import ctypes as C
    lib = _lib.load()
    fn_on, fn_read = lib.mtmc_dbg_panel_timing, lib.mtmc_dbg_panel_times
    fn_read.argtypes = [C.POINTER(C.c_float), C.c_int32]
    if fn_on(1) != 0:
        return None
    tot, n_panels = 0.0, 0
    buf = (C.c_float * 32)()
    try:
        with torch.no_grad():
            for it in range(reps + 1):
                model(data)
                n = fn_read(buf, 32)
                if n == 0:
                    return None
                if it > 0:
                    tot += sum(buf[i] for i in range(n))
                    n_panels = n
    finally:
        fn_on(0)
    return tot / reps, n_panels


def run_single(name, device, steps, warmup, with_cpu=True, phase_iters=20, legs=True):
    desc, L, cs = WORKLOADS[name]
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=cs)
    torch.manual_seed(0)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).to(device).eval()
    model.deterministic = bool(os.environ.get("MTMC_DETERMINISTIC"))
    data = make_workload(name, device)
    n, e = data.x.shape[0], data.edge_index.shape[1]
    sec, dist_ms = time_forward(model, data, steps, warmup)
    launch_mode, eager_ms, replay_ms = "eager launches", sec * 1e3, None
    if e <= 2_000_000 and not os.environ.get("MTMC_NO_GRAPH"):
        # few-edge graphs: ~20 launches per forward can cost the host more than the forward costs the GPU; the same
        # forward replayed from a HIP graph (model.capture) is immune to that -- report the faster way of launching it
        with torch.no_grad():
            replay = model.capture(data)
        r_sec, r_dist = time_forward(model, data, steps, warmup, fn=replay)
        replay_ms = r_sec * 1e3
        if replay_ms < eager_ms:
            sec, dist_ms, launch_mode = r_sec, r_dist, "HIP graph replay (model.capture)"
    seq, ms = time_phases(model, data, phase_iters)
    spec = model.spec
    # dominant kernel = the kernel NAME with the largest summed time (all its launches in a step, the same
    # granularity as a rocprofv3 --stats row); phases that launch nothing (un-split combine) are skipped
    enc_names = encoder_kernels(model, n, e)
    from mtmc_mpn import engine as _engine
    plan = (model._engine or _engine.ForwardEngine(model)).plan(n, e)     # which kernels the library runs for this size

    # pass C and node_proj have several kernels behind one phase: the symbol rocprofv3 lists, for the counter summaries
    symbols = {"pass_c_kernel": {_lib.PASS_C_MFMA_ANY: "pass_c_mfma_kernel", _lib.PASS_C_MFMA_SORTED: "pass_c_sorted_kernel"}
               .get(plan.pass_c, "pass_c_kernel"), "node_proj_kernel": "node_proj_mfma_kernel"}

    def kernel_of(ph, arg):
        if ph != _lib.PH_NODE_ENC:
            return PHASE_NAMES[ph]
        # the few-row later layers are different instantiations per layer (rocprofv3 lists them separately)
        return enc_names[arg] + (f"[layer {arg}]" if enc_names[arg] == "few_wave_kernel" else "")
    launched = [(pa, t) for pa, t in zip(seq, ms) if not (pa[0] == _lib.PH_NODE_COMBINE and plan.enc_split_k[pa[1]] <= 1)]
    by_kind = {}
    for (ph, arg), t in launched:
        by_kind.setdefault(kernel_of(ph, arg), []).append(((ph, arg), t))
    dom_key = max(by_kind, key=lambda k: sum(t for _, t in by_kind[k]))

    def roofline_of(key):
        launches = by_kind[key]
        avg_ms = sum(t for _, t in launches) / len(launches)
        kinds = [phase_cost(ph, arg, spec, n, e) for (ph, arg), _ in launches]
        bound = kinds[0][0]
        work = sum(w for _, w in kinds) / len(kinds)
        peak_note = None
        base = key.split("[")[0]
        if bound == "mfma" and base in ("gemm_bn_f16x3_kernel", "gemm_f16p_m16_kernel", "gemm_staged_kernel", "gemm_rows_kernel",
                                        "few_l0_kernel", "few_wave_kernel"):
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS / 3.0, "TFLOP/s"
            peak_note = ("algorithmic fp32 flops (2*M*N*K) against the fp16 dense MFMA peak (= the bf16 one) / 3: the kernel "
                         "reaches fp32 accuracy with three fp16 products per fp32 product")
        elif bound == "mfma" and base == "gemm_bn_bf16x6_kernel":
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_BF16_PEAK_TFLOPS / 6.0, "TFLOP/s"
            peak_note = ("algorithmic fp32 flops (2*M*N*K) against the bf16 dense MFMA peak / 6: the kernel reaches fp32 "
                         "accuracy with six bf16 products per fp32 product")
        elif bound == "mfma":
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        rl = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
              "traffic": None, "kernel": key, "avg_kernel_ms": avg_ms, "launches_per_step": len(launches),
              "algorithmic_per_launch": work, "timed": "HIP events around each launch of the phase-by-phase forward"}
        if peak_note:
            rl["peak_note"] = peak_note
        return rl
    roofline = roofline_of(dom_key)
    roofline["event_pair_floor_ms"] = empty_event_pair_ms()
    if dom_key == "gemm_f16p_m16_kernel" and plan.layer0_panels > 1:
        # the one-call forward (what ms_per_step times) runs this layer as row panels beside the operand split: report THAT
        # structure -- the sum of its panel launches -- and keep the phase path's single launch as a second, named figure
        pt = panel_times(model, data)
        if pt is not None:
            per_step_ms, n_panels = pt
            work_step = roofline["algorithmic_per_launch"]
            single = {k: roofline[k] for k in ("achieved", "frac", "avg_kernel_ms", "launches_per_step", "timed")}
            roofline.update({"achieved": work_step / (per_step_ms * 1e-3) / 1e12, "kernel_ms_per_step": per_step_ms,
                             "avg_kernel_ms": per_step_ms / n_panels, "launches_per_step": n_panels,
                             "algorithmic_per_launch": work_step / n_panels, "algorithmic_per_step": work_step,
                             "timed": "HIP events the library records around every panel launch of the ONE-CALL forward "
                                      "(mtmc_dbg_panel_timing): the structure ms_per_step times; the operand split of the next "
                                      "panel runs beside each launch on a side stream",
                             "phase_path_single_launch": single})
            roofline["frac"] = roofline["achieved"] / roofline["peak"]
    dom_symbol = symbols.get(dom_key.split("[")[0], dom_key.split("[")[0])
    roofline["kernel_symbol"] = dom_symbol
    roofline["traffic"] = pmc_traffic(name, dom_symbol, roofline["launches_per_step"])
    if roofline["traffic"] is not None:
        roofline["traffic_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/%s_%s_pmc_*.txt "
                                    "(separate rocprofv3 --pmc passes; gfx950 half-count correction on reads)"
                                    % (pmc_round(name), name))
    # the single longest launch, when it is not the kernel with the largest summed time (the few-row layer 0 on the headline graph)
    longest = max((it for it in launched if it[0][0] != _lib.PH_BEGIN), key=lambda it: it[1])   # (PH_BEGIN = memset + prep_kernel: two operations)
    longest_key = kernel_of(*longest[0])
    b_fwd = algorithmic_bytes_forward(n, e, L, cs)
    phases = {}
    for (ph, arg), t in launched:
        k = kernel_of(ph, arg) + (f"[{arg}]" if ph == _lib.PH_NODE_ENC and "[" not in kernel_of(ph, arg) else "")
        phases[k] = phases.get(k, 0.0) + t
    res = {"workload": name, "description": desc, "N": n, "E": e, "L": L, "Cs": cs,
           "value": e / sec, "ms_per_step": sec * 1e3, "edge_rounds_per_s": e * L / sec,
           "launch": {"mode": launch_mode, "eager_ms": eager_ms, "graph_replay_ms": replay_ms},
           "step_ms": {k: round(v, 5) for k, v in dist_ms.items()},
           "roofline": roofline,
           "forward_algorithmic": {"bytes": b_fwd, "GBps": b_fwd / sec / 1e9, "frac_of_hbm_peak": b_fwd / sec / 1e9 / HBM_PEAK_GBS,
                                   "note": "SURVEY 8(d) reference-formulation bytes / whole-forward time"},
           "phase_ms_sum": sum(t for _, t in launched), "phase_ms": {k: round(v, 4) for k, v in phases.items()},
           "plan": {"encoder_kernels": enc_names, "enc_split_k": plan.enc_split_k, "edges_per_thread": plan.edges_per_thread,
                    "lazy_edges": plan.lazy_edges, "pass_c": ["walk", "mfma_sorted", "mfma_any"][plan.pass_c],
                    "pass_a_col_blocks": plan.pass_a_col_blocks, "layer0_panels": plan.layer0_panels,
                    "enc2_passenger": plan.enc2_passenger, "node_stat_folded": plan.node_stat_folded,
                    "layer0_pipeline": os.environ.get("MTMC_L0_PIPELINE", "1 (default: row panels, split on a side stream)")},
           # what every timed forward does about the weights: nothing is trusted from one forward to the next
           "weight_plane_cache": {"enabled": bool(model.cache_weight_planes),
                                  "validity": "verified on the device in EVERY timed forward: 64-bit fingerprint per 8 weight rows "
                                              "against the fp32 weights (csrc/split_body.h); no host-side promise"}}
    if longest_key != dom_key:
        res["roofline_longest_launch"] = roofline_of(longest_key)
    if legs:
        # the same workload WITHOUT the weight-plane cache (every weight-derived operand made again per forward; few-row graphs:
        # the split-K kernels of rounds 1-4)
        model.cache_weight_planes = False
        u_sec, _ = time_forward(model, data, max(5, steps // 2), max(2, warmup // 2))
        model.cache_weight_planes = True
        res["ms_per_step_uncached"] = u_sec * 1e3
    if with_cpu:
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        res["cpu_baseline"] = cpu_baseline(name, params, sd, data)
    del data, model
    torch.cuda.empty_cache()
    return res


def exact_fp32_leg(name, steps, warmup):
    """ms_per_step of the same workload with the exact-fp32 MFMA encoder (MTMC_GEMM_FP32=1: v_mfma_f32_32x32x2_f32 instead of
    the three-product fp16 split).  The switch is read once per process: a child process of this script."""
    import subprocess
    env = dict(os.environ, MTMC_GEMM_FP32="1", MTMC_BENCH_CHILD="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu", "--no-stress"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception as ex:   # noqa: BLE001  (a diagnostic leg must not take the bench line down)
        return f"failed: {ex}"


def run_graph_build(device, with_cpu=True):
    """SURVEY 8(f)-1,2: raw tracklet features -> (x, edge_index, edge_attr, labels), S02 ground-truth topology."""
    import numpy as np
    cams = np.repeat(np.arange(4), graphs.S02_GT_CAMS)
    feats = torch.randn(cams.size, 2048, generator=torch.Generator().manual_seed(2))
    labels = torch.randint(0, 145, (cams.size,), generator=torch.Generator().manual_seed(3)).numpy()
    fg = feats.to(device)
    for _ in range(5):
        g = mtmc_mpn.build_graph(fg, cams, labels)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 30
    for _ in range(reps):
        g = mtmc_mpn.build_graph(fg, cams, labels)
    torch.cuda.synchronize(device)
    sec = (time.perf_counter() - t0) / reps
    e = g.edge_index.shape[1]
    out = {"workload": "S02 ground-truth topology from raw features (normalise, edges, labels, edge_attr)", "N": int(cams.size),
           "E": e, "ms_per_build": sec * 1e3, "edges_per_s": e / sec,
           "replaces": "inference.py:402-456 (16 KB of 2048-d gathers per edge -> one Gram matrix + 8 B/edge)"}
    if with_cpu:
        from oracle import graph_oracle
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        graph_oracle.build(feats, cams, labels)
        out["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    return out


def run_postprocess(device, with_cpu=True):
    """last logits -> (predictions, ID_pred), S02-scale scenario (the pp8 fixture's recipe)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pp_cases          # seeded post-processing scenarios (test infrastructure, shared with the fixtures)
    kw = dict(n_ids=140, n_cams=4, seed=20, fp_rate=0.0005, fn_rate=0.05, pair_fp=0.0006)
    sc = pp_cases.scenario(**kw)
    logits, ei = sc.logits.to(device), sc.edge_index.to(device)
    for _ in range(3):
        out = mtmc_mpn.postprocess(logits, ei, sc.n_nodes, sc.n_cams)
    info = out.info
    torch.cuda.synchronize(device)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = mtmc_mpn.postprocess(logits, ei, sc.n_nodes, sc.n_cams, check=False)
    torch.cuda.synchronize(device)
    sec = (time.perf_counter() - t0) / reps
    # the same graph with a cleaner classifier (few over-sized clusters): what a trained model's output looks like
    sc2 = pp_cases.scenario(**dict(kw, fp_rate=0.0002, pair_fp=0.00005))
    l2, e2 = sc2.logits.to(device), sc2.edge_index.to(device)
    info2 = mtmc_mpn.postprocess(l2, e2, sc2.n_nodes, sc2.n_cams).info
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        mtmc_mpn.postprocess(l2, e2, sc2.n_nodes, sc2.n_cams, check=False)
    torch.cuda.synchronize(device)
    sec2 = (time.perf_counter() - t0) / reps
    res = {"workload": "S02-scale synthetic logits (pp8 fixture recipe): softmax/argmax, cut, pruning, cut, splitting, SCC",
           "N": sc.n_nodes, "E": int(ei.shape[1]), "ms_per_call": sec * 1e3, "edges_per_s": ei.shape[1] / sec, **info,
           "clean_case": {"ms_per_call": sec2 * 1e3, **info2},
           "replaces": "inference.py:475-489 + post_processing :70-169 + utils.py:30-339 (D2H of the edge list, Python "
                       "list scans, networkx SCC per splitting iteration)"}
    if with_cpu:
        from oracle import postprocess_oracle as po
        t0 = time.perf_counter()
        prob, pred = po.classify(sc.logits)
        po.post_processing(sc.n_cams, pred, sc.edge_index, sc.n_nodes, prob)
        res["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        res["cpu_note"] = ("oracle = the reference's algorithm with set/array lookups instead of its O(A^2) list scans; the "
                           "reference's own post_processing took 9.3 s on this case in the build container (8 cores)")
    return res


def training_setup(device):
    """BASELINE config 3 shape: 100 identities of the training scenes (N~430, E~173k), L=3, Cs=3; SGD lr 0.01, momentum 0.9, wd 1e-4."""
    import json as _json
    with open(os.path.join(ROOT, "tests", "golden", "train_tracklets.json")) as f:
        tr = _json.load(f)["tracklets"]
    d = graphs.training_graph(tr, 100, 2048, 3)
    params = mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=3)
    torch.manual_seed(0)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).to(device).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)   # one kernel for all 34 tensors
    ei = d.edge_index.t().contiguous().to(device).t()
    data = types.SimpleNamespace(x=d.x.to(device), edge_index=ei, edge_attr=d.edge_attr.to(device))
    labels = d.edge_labels.long().to(device)
    # the shipped loss (config_training.yaml: CE_weighted, loss_weight_custom): class weights 1 and n0/n1 per batch,
    # sum(w*l)/sum(w) (train.py:118-138) = weighted mean cross-entropy; its FPR term carries no gradient and is left out
    n1 = float(labels.sum())
    ce_weight = torch.tensor([1.0, (labels.numel() - n1) / max(n1, 1.0)], device=device)
    return model, opt, data, labels, ce_weight


def training_graph_child(device, reps=60):
    """The config-3 training step as ONE HIP graph (MOTMPNet.capture_training_step: Dropout seed from a device counter): the
    host's launch time drops out.  Run as a child process of the bench (a failed capture can take a ROCm process down)."""
    model, opt, data, labels, ce_weight = training_setup(device)
    replay = model.capture_training_step(
        data, lambda o, _h: mtmc_mpn.cross_entropy_steps(o["classified_edges"], labels, weight=ce_weight), opt)
    for _ in range(20):
        replay()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        gl = replay()
    torch.cuda.synchronize(device)
    gsec = (time.perf_counter() - t0) / reps
    e = data.edge_index.shape[1]
    return {"ms_per_step": gsec * 1e3, "edges_per_s": e / gsec, "final_loss": float(gl.detach()), "steps_before": 23 + reps,
            "note": "forward + loss + backward + fused SGD captured once, replayed; new Dropout masks per replay"}


def training_graph_leg():
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train-graph-child"]
    try:
        r = subprocess.run(cmd, env=dict(os.environ, MTMC_BENCH_CHILD="1"), capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:   # noqa: BLE001  (a diagnostic leg must not take the bench line down)
        return {"error": repr(ex)[:300]}


def run_training_step(device, graph_leg=True):
    """forward with Dropout + class-weighted cross-entropy over the classified steps + backward + SGD step, launch by launch;
    `graph_replay`: the same step as one HIP graph (child process)."""
    model, opt, data, labels, ce_weight = training_setup(device)

    def step():
        opt.zero_grad(set_to_none=True)
        out, _ = model(data)
        loss = mtmc_mpn.cross_entropy_steps(out["classified_edges"], labels, weight=ce_weight)   # = the sum over the steps
        loss.backward()
        opt.step()
        return loss
    import gc
    gc.collect()
    for _ in range(20):
        step()
    torch.cuda.synchronize(device)
    reps = 60
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    host = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(reps):
        h0 = time.perf_counter()
        loss = step()
        host.append((time.perf_counter() - h0) * 1e3)
        marks[i + 1].record()
    torch.cuda.synchronize(device)
    sec = (time.perf_counter() - t0) / reps
    gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(reps)]
    e = data.edge_index.shape[1]
    out = {"workload": "training-scene topology, 100 identities: forward(Dropout) + class-weighted CE (the shipped loss) + backward + SGD, L=3 Cs=3",
           "N": int(data.x.shape[0]), "E": int(e), "ms_per_step": sec * 1e3, "edges_per_s": e / sec,
           "step_ms_gpu_side": {k: round(v, 4) for k, v in percentiles(gpu).items()},
           "step_ms_host_issue": {k: round(v, 4) for k, v in percentiles(host).items()},
           "final_loss": float(loss.detach())}
    del loss, model, opt
    out["ms_per_step_eager"], out["launch"] = out["ms_per_step"], "eager launches"
    if graph_leg:
        g = out["graph_replay"] = training_graph_leg()
        if "ms_per_step" in g and g["ms_per_step"] < out["ms_per_step"]:
            out["ms_per_step"], out["edges_per_s"] = g["ms_per_step"], g["edges_per_s"]
            out["launch"] = "HIP graph replay (MOTMPNet.capture_training_step)"
    kt = train_kernel_ms()
    if kt is not None:
        out["kernel_ms_per_step"] = kt
        out["kernel_note"] = ("sum of kernel durations per step from the committed rocprofv3 --kernel-trace --stats summary "
                              "(profiles/%s): what the GPU needs; the rest of ms_per_step is host launch time" % kt["source"])
    return out


def train_kernel_ms():
    """Kernel time of one training step from the newest committed rocprofv3 summary (tools/train_profile.sh)."""
    import csv
    for rnd in committed_rounds():
        path = os.path.join(ROOT, "profiles", f"{rnd}_train_kernel_stats.csv")
        meta = os.path.join(ROOT, "profiles", f"{rnd}_train_kernel_stats.steps")
        if os.path.exists(path):
            steps = int(open(meta).read().split()[0]) if os.path.exists(meta) else None
            total_ns = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(path)) if r.get("TotalDurationNs"))
            if steps:
                return {"ms": total_ns / steps * 1e-6, "source": os.path.basename(path), "profiled_steps": steps}
    return None


def add_measured_peak(res, probe):
    """Fractions against the bandwidth this box actually delivers (hbm_probe), beside the ones against the 8 TB/s spec."""
    peak = probe["measured_peak_GBps"]
    res["forward_algorithmic"]["frac_of_measured_peak"] = res["forward_algorithmic"]["GBps"] / peak
    if res["roofline"]["bound"] == "hbm":
        res["roofline"]["frac_of_measured_peak"] = res["roofline"]["achieved"] / peak


def self_launch_command(args, port=None):
    """`python bench.py --gpus N` WITHOUT a launcher (no RANK / WORLD_SIZE in the environment): the command that starts
    the N ranks, exactly as the driver would -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload]
    if args.no_stress:
        cmd.append("--no-stress")
    if args.no_cpu:
        cmd.append("--no-cpu")
    return cmd


def launch_or_refuse(args):
    """Called before anything touches a GPU.  Returns None when this process should run the bench itself, else the exit
    code of the ranks it started (or 2 after printing why it cannot).  A `--gpus N` process never sits in a rendezvous
    waiting for ranks nobody started."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None or "RANK" in os.environ:
        world = int(env_world or "1")
        if world != args.gpus and not (args.gpus == 1 and world >= 1):
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            return 2
        return None                                         # a rank of a launcher's job (torchrun): run
    if args.gpus <= 1:
        return None
    import subprocess
    have = torch.cuda.device_count()                        # (counting devices does not initialise the GPU)
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible -- not starting a job that cannot "
              "rendezvous", file=sys.stderr)
        return 2
    cmd = self_launch_command(args)
    print("bench.py: no launcher environment (RANK / WORLD_SIZE); starting the ranks: " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd).returncode                   # children inherit stdout: rank 0's JSON line is ours


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--no-stress", action="store_true", help="skip the config-4 stress graph in the 1-GPU run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--train-graph-child", action="store_true", help=argparse.SUPPRESS)   # (training_graph_leg's child)
    args = ap.parse_args()
    if args.train_graph_child:
        torch.cuda.set_device(0)
        print(json.dumps(training_graph_child(torch.device("cuda:0"))))
        return

    rc = launch_or_refuse(args)
    if rc is not None:
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("MTMC_BENCH_FORCE_DIST"):
        from bench_dist import main_distributed   # multi-GPU leg lives beside this file
        return main_distributed(args)

    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    name = "s02" if args.workload == "auto" else args.workload
    probe = hbm_probe(device)
    child = bool(os.environ.get("MTMC_BENCH_CHILD"))
    res = run_single(name, device, args.steps, args.warmup, with_cpu=not args.no_cpu, legs=not child)
    add_measured_peak(res, probe)
    line = {"metric": "MPN forward edges/sec (+ achieved roofline fraction of the dominant kernel)",
            "value": res["value"], "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded features on the reference's S02 ground-truth topology; random-init weights)",
            "config": {"workload": f"{name}: {res['description']}, L={res['L']}, Cs={res['Cs']}, eval forward",
                       "N": res["N"], "E": res["E"], "parallelism": "1 GPU"},
            "roofline": res["roofline"], "cpu_baseline": res.get("cpu_baseline"), "launch": res["launch"],
            "forward_algorithmic": res["forward_algorithmic"], "edge_rounds_per_s": res["edge_rounds_per_s"],
            "phase_ms": res["phase_ms"], "step_ms": res["step_ms"], "hbm_probe": probe, "plan": res["plan"],
            "weight_plane_cache": res["weight_plane_cache"], "ms_per_step_uncached": res.get("ms_per_step_uncached")}
    if "roofline_longest_launch" in res:
        line["roofline_longest_launch"] = res["roofline_longest_launch"]
    if not child and not args.no_stress:
        line["ms_per_step_exact_fp32"] = exact_fp32_leg(name, args.steps, args.warmup)
    if args.workload == "auto" and not args.no_stress:
        # (20 timed forwards after 5: with 5 after 2 the first clock ramp of a 2.6 ms forward was part of the figure)
        st = run_single("cfg4", device, max(20, args.steps // 5), max(5, args.warmup // 4), with_cpu=not args.no_cpu,
                        phase_iters=5)
        add_measured_peak(st, probe)
        line["stress"] = st
        # the multi-GPU workload (config 5) on this one GPU: the N=1 point of the strong-scaling curve
        line["scale_base"] = run_single("cfg5", device, 6, 3, with_cpu=False, phase_iters=2)
        add_measured_peak(line["scale_base"], probe)
        line["scale_base"]["note"] = ("1-GPU point of the strong-scaling series that `--gpus N` (N > 1) reports: the same "
                                      "1M-node / 100M-edge graph, one call, no collectives")
        # the two config-2 legs BASELINE.json words: the shipped L = 1 model on the S02 topology, and the tracker-scale graph
        line["extra"] = {}
        for leg in ("s02_L1", "s02_tracker"):
            r = run_single(leg, device, args.steps, args.warmup, with_cpu=False, phase_iters=5)
            add_measured_peak(r, probe)
            line["extra"][leg] = {k: r[k] for k in ("description", "N", "E", "L", "Cs", "value", "ms_per_step", "ms_per_step_uncached",
                                                     "edge_rounds_per_s", "launch", "step_ms", "roofline", "phase_ms", "plan")}
        line["graph_build"] = run_graph_build(device, with_cpu=not args.no_cpu)
        line["training_step"] = run_training_step(device)
        line["postprocess"] = run_postprocess(device, with_cpu=not args.no_cpu)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
