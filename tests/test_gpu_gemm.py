"""The node encoder's GEMM dispatch (fp16 two-piece kernel, every tile configuration) against fp64."""
import pytest
import torch

from mtmc_mpn import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(A, W, b, with_stats=True):
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    Y = torch.empty(M, N, device=DEV)
    scr = torch.zeros(48, dtype=torch.int32, device=DEV)
    st = torch.empty(2 * N, dtype=torch.float64, device=DEV) if with_stats else None
    _lib.check(lib.mtmc_linear_raw(A.data_ptr(), A.stride(0), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N,
                                   scr.data_ptr(), st.data_ptr() if st is not None else None,
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return Y, st


# (M, K, N): split-K 64x64 tiles | un-split 64x64 | 128x128 tiles (several resident per CU) | BK=32 | ragged edges
SHAPES = [(450, 2048, 1024), (450, 128, 32), (3000, 2048, 1024), (9000, 2048, 1024), (20011, 1024, 512),
          (9000, 96, 1024), (777, 160, 200), (33, 32, 7), (9000, 2048, 130)]


@pytest.mark.parametrize("shape", SHAPES)
def test_against_fp64(shape):
    M, K, N = shape
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    ref = A.double() @ W.double().t() + b.double()
    for rep in range(2):                                   # twice: results must not depend on what ran before
        Y, st = run(A, W, b)
        err = float((Y.double() - ref).abs().max())
        assert err <= 2e-6 * float(ref.abs().max()) * max(1.0, (K / 512) ** 0.5), (shape, rep, err)
        assert torch.allclose(st[:N], ref.sum(0), rtol=0, atol=1e-5 * float(ref.abs().sum(0).max()))
        assert torch.allclose(st[N:], (ref * ref).sum(0), rtol=1e-5, atol=0)


@pytest.mark.parametrize("scale_a, scale_w", [(1e-4, 1.0), (300.0, 1e-3), (1.0, 40.0), (1e-25, 1e-8), (1e12, 1e10)])
def test_operand_ranges(scale_a, scale_w):
    """The power-of-two operand scaling keeps tiny and large inputs inside fp16's range."""
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(2000, 512, generator=g) * scale_a).to(DEV)
    A[3, 7] = 50 * scale_a                                  # one outlier sets the scale; the rest must survive it
    W = (torch.randn(256, 512, generator=g) * scale_w / 22).to(DEV)
    b = torch.zeros(256, device=DEV)
    ref = A.double() @ W.double().t()
    Y, _ = run(A, W, b, with_stats=False)
    assert float((Y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


def test_zero_and_strided_input():
    A = torch.zeros(100, 64, device=DEV)
    W = torch.randn(40, 64, device=DEV)
    b = torch.arange(40, device=DEV, dtype=torch.float32)
    Y, _ = run(A, W, b)
    assert torch.equal(Y, b.expand(100, 40))
    big = torch.randn(300, 256, device=DEV)
    A2 = big[:, :128]                                        # row stride 256, 128 columns used
    Y2, _ = run(A2, torch.randn(64, 128, device=DEV, generator=None) * 0 + 1.0, torch.zeros(64, device=DEV))
    assert torch.allclose(Y2, A2.sum(1, keepdim=True).expand(300, 64), atol=1e-4)
