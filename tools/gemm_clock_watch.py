#!/usr/bin/env python3
"""Runs the layer-0 pre-split GEMM (100000 x 2048 x 1024) back to back for a few seconds on random and on all-zero
operands while a thread samples the shader clock and the socket power the driver reports (rocm-smi, read-only):
    python tools/gemm_clock_watch.py
Same instruction stream both times; what differs is the clock the power management grants (point 4)."""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtmc_mpn import _lib  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import lab_lib  # noqa: E402  (the kernel laboratory's loader lives with the tests)

lib = _lib.load()
M, K, N = 100000, 2048, 1024
s = torch.cuda.current_stream().cuda_stream


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as exc:                                   # no rocm-smi / no permission: report and stop sampling
            out.append(("error", str(exc)))
            return
        sclk = re.findall(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", txt)
        pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", txt)
        out.append((int(sclk[0]) if sclk else None, float(pw[0]) if pw else None))
        time.sleep(0.2)


def run(kind, seconds=4.0):
    A = torch.randn(M, K, device="cuda")
    W = (torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5
    if kind == "zeros":
        A.zero_(), W.zero_()
    b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(M * K * 4 + N * K * 4 + (M + N) * 4 + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")

    def gemm(variant):
        rc = lab_lib.load_lab().mtmc_lab_linear_presplit_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N,
                                                          work.data_ptr(), work.numel(), scr.data_ptr(), st.data_ptr(), variant, s)
        assert rc == 0, rc
    gemm(9)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            gemm(-10)                                              # GEMM alone on the planes already in `work`
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    clk = [c for c, _ in samples if isinstance(c, int)]
    pw = [p for _, p in samples if isinstance(p, float)]
    print(f"{kind:6s}: {ms * 1e3:7.1f} us per GEMM over {n} launches; sclk samples (MHz) {clk}; power samples (W) {pw}"
          + (f"; sampler: {samples[0]}" if samples and samples[0][0] == "error" else ""), flush=True)


run("randn")
run("zeros")
run("randn")
