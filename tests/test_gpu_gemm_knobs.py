"""The older encoder kernels stay selectable (MTMC_GEMM_NO_F16: bf16 three-piece split, MTMC_GEMM_FP32: exact fp32 MFMA);
the knobs are read once per process, so each runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from mtmc_mpn import _lib
lib = _lib.load()
worst = 0.0
for (M, K, N) in [(450, 2048, 1024), (9000, 1024, 512), (777, 160, 200)]:
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g).cuda()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
    b = torch.zeros(N).cuda()
    Y = torch.empty(M, N, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    _lib.check(lib.mtmc_linear_raw(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, scr.data_ptr(), None,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t()
    worst = max(worst, float((Y.double() - ref).abs().max() / ref.abs().max()))
print("WORST", worst)
assert worst <= 3e-6, worst
""" % ROOT


@pytest.mark.parametrize("knob", ["MTMC_GEMM_NO_F16", "MTMC_GEMM_FP32"])
def test_older_kernels_still_exact(knob):
    env = dict(os.environ, **{knob: "1"})
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "WORST" in r.stdout
