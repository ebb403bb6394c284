"""The few-row encoder kernels behind mtmc_linear_few_raw (csrc/gemm_few.hip): layer 0 on pre-split operands (64 x 32 tiles
over all of K, LDS-DMA ring) and the later layers with K cut between the waves of a workgroup, against float64 -- ragged row
counts (1 .. 1100: partial 16- and 64-row tiles, a single row), every wave count 1 .. 8, both column-block widths, strided A,
columns BatchNorm kills or keeps entirely, rows of very different magnitude.  Same error budget as the many-row kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(A, W, b, st_in=None, gamma=None, beta=None, count=None):
    from mtmc_mpn import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    Y = torch.full((M, N), float("nan"), device="cuda")
    work = torch.empty(4 * M * K + 4 * N * K + 4 * (M + N) + 1024, dtype=torch.uint8, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.mtmc_linear_few_raw(A.data_ptr(), A.stride(0), p(st_in), p(gamma), p(beta), float(count or M), W.data_ptr(),
                                       b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(), st.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return Y, st


def _check_stats(Y, st, M):
    N = Y.shape[1]
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * max(1.0, Y.abs().max().item()) * M)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-9, atol=1e-30)


@pytest.mark.parametrize("shape", [(450, 2048, 1024), (1002, 2048, 1024), (64, 2048, 1024), (9, 2048, 1024), (1, 64, 32), (17, 64, 32),
                                   (65, 128, 64), (300, 192, 96), (450, 2048, 32), (130, 1024, 256), (1100, 2048, 1024)])
def test_layer0_matches_float64(shape):
    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, device="cuda", generator=g) * (0.05 + torch.rand(M, 1, device="cuda", generator=g))   # rows of different scale
    if M > 3:
        A[3] *= 1e-4
        A[M // 2, K // 3] = 40.0                                # an outlier sets its row's scale
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    W[N // 3] *= 1e-3
    b = torch.randn(N, device="cuda", generator=g)
    Y, st = _run(A, W, b)
    ref = A.double() @ W.double().t() + b.double()
    bound = 3e-7 * (A.double().abs() @ W.double().abs().t() + b.double().abs())
    assert torch.isfinite(Y).all()
    err = ((Y.double() - ref).abs() / bound).max().item()
    assert err < 1.0, err
    _check_stats(Y, st, M)


def test_layer0_strided_rows():
    g = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(200, 2048 + 64, device="cuda", generator=g)
    A = big[:, :2048]                                           # row stride 2112 floats
    W = (torch.rand(64, 2048, device="cuda", generator=g) * 2 - 1) / 45
    b = torch.zeros(64, device="cuda")
    Y, _ = _run(A, W, b)
    ref = A.double() @ W.double().t()
    assert ((Y.double() - ref).abs() / (3e-7 * (A.double().abs() @ W.double().abs().t()) + 1e-30)).max().item() < 1.0


def _ref_bn(A, st_in, gamma, beta, W, b, count, split):
    K = A.shape[1]
    mean = st_in[:K] / count
    var = (st_in[K:] / count - mean * mean).clamp_min(0)
    s = torch.rsqrt(var + 1e-5) * gamma.double()
    t = beta.double() - mean * s
    a = torch.relu(A.double() * s + t)
    da = (A.double().abs() * s.abs() + t.abs()) * 2.0 ** -23
    bound = split * (a.abs() @ W.double().abs().t() + b.double().abs()) + da @ W.double().abs().t()
    return a, a @ W.double().t() + b.double(), bound


# (M, K, N): K / 128 in 2..8 -> four k-steps per wave (2..8 waves); K / 32 <= 8 otherwise -> one k-step per wave (1..8 waves);
# K = 2048 -> eight k-steps per wave; N % 32 == 0 with >= 256 workgroups -> two column blocks per workgroup, else one
@pytest.mark.parametrize("shape", [(450, 1024, 512), (450, 512, 128), (450, 128, 32), (1002, 1024, 512), (1002, 512, 128),
                                   (9, 1024, 512), (1, 128, 32), (33, 32, 16), (100, 96, 48), (100, 64, 16), (450, 256, 32),
                                   (450, 768, 96), (130, 2048, 64), (300, 224, 32), (2000, 384, 64), (450, 640, 512)])
def test_later_layers_match_float64(shape):
    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, device="cuda", generator=g) * (1 + 4 * torch.rand(1, K, device="cuda", generator=g)) + \
        torch.randn(1, K, device="cuda", generator=g)
    gamma = 0.5 + torch.rand(K, device="cuda", generator=g)
    beta = 0.2 * torch.randn(K, device="cuda", generator=g)
    beta[0] = -50.0                                           # a column BatchNorm + ReLU kills entirely
    beta[1] = 30.0                                            # ... and one that sets its wave's operand scale
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    W[N // 3] *= 1e-3
    b = torch.randn(N, device="cuda", generator=g)
    count = float(max(M, 2))
    st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    if M == 1:                                                # (a single row of a larger batch: statistics of 2 rows)
        st_in = torch.cat([2 * A.double().sum(0) + 1, 2 * (A.double() ** 2).sum(0) + 3]).contiguous()
    Y, st = _run(A, W, b, st_in, gamma, beta, count)
    a, ref, bound = _ref_bn(A, st_in, gamma, beta, W, b, count, split=2.0 ** -19 if K < 512 else 5e-7)
    assert torch.isfinite(Y).all()
    if M > 1:
        assert (a[:, 0] == 0).all()
    err = ((Y.double() - ref).abs() / bound).max().item()
    assert err < 1.0, err
    _check_stats(Y, st, M)


def test_later_layer_is_reproducible():
    """The waves' partial tiles are added in a fixed order: bitwise equal from run to run."""
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(450, 1024, device="cuda", generator=g)
    gamma, beta = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
    W = torch.randn(512, 1024, device="cuda", generator=g) / 32
    b = torch.zeros(512, device="cuda")
    st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    Y0, _ = _run(A, W, b, st_in, gamma, beta)
    for _ in range(3):
        Y1, _ = _run(A, W, b, st_in, gamma, beta)
        assert torch.equal(Y0, Y1)


def test_refuses_shapes_it_does_not_take():
    from mtmc_mpn import _lib
    lib = _lib.load()
    A = torch.zeros(10, 48, device="cuda")
    W = torch.zeros(16, 48, device="cuda")
    b = torch.zeros(16, device="cuda")
    Y = torch.zeros(10, 16, device="cuda")
    work = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = lib.mtmc_linear_few_raw(A.data_ptr(), 48, None, None, None, 10.0, W.data_ptr(), b.data_ptr(), Y.data_ptr(), 10, 48, 16,
                                 work.data_ptr(), work.numel(), None, None)
    assert rc == _lib.E_ARG                                   # layer 0 wants K % 64 == 0 and N % 32 == 0
