"""The many-row encoder GEMMs behind mtmc_linear_staged_raw -- the role-split kernel (csrc/gemm_staged.hip, N % 256 == 0) and
the row-streaming kernel for the narrow last layer (csrc/gemm_rows.hip, 128 -> 32): Y = relu(bn(A)) . W^T + b against float64 --
ragged row counts (tile heights 80..128 are picked per launch), odd k-tile counts, strided A, columns whose
BatchNorm kills or keeps everything, an outlier element.  Bound: 5e-7 * (sum_k |a||w| + |b|) plus
the fp32 rounding of the BatchNorm affine itself (see _ref)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(A, gamma, beta, W, b, count=None, lab=False):
    """lab=True: the kernel laboratory's second form of the role-split kernel (csrc/lab/staged2_lab.hip, libmtmc_lab.so:
    W fragments straight into registers, four A stages, one barrier per two k-tiles) through the same argument list."""
    from mtmc_mpn import _lib
    lib = _lib.load()
    M, K = A.shape
    N = W.shape[0]
    count = float(M if count is None else count)
    st_in = torch.cat([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    Y = torch.full((M, N), float("nan"), device="cuda")
    work = torch.empty(4 * N * K + 4 * N + 512, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    args = (A.data_ptr(), A.stride(0), st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), count,
            W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(),
            scr.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if lab:
        import lab_lib
        assert lab_lib.load_lab().mtmc_lab_linear_staged2_raw(*args) == 0
    else:
        _lib.check(lib.mtmc_linear_staged_raw(*args))
    torch.cuda.synchronize()
    return Y, st, scr[32:48].view(torch.float32).max().item(), st_in


def _ref(A, st_in, gamma, beta, W, b, count, split=5e-7):
    """float64 reference and the error budget per output: 5e-7 * (sum_k |a||w| + |b|) for the operand split (the other
    split kernels' tests use 3e-7; with as few as 64 terms and truncating conversions the worst element gets closer to the
    2^-20 worst case of a two-piece rtz split) + what evaluating the affine s*y + t itself in fp32 may cost (the reference
    path does that in fp32 too): two roundings of relative size 2^-24 on |s*y| + |t| per element, carried through |w|."""
    K = A.shape[1]
    mean = st_in[:K] / count
    var = (st_in[K:] / count - mean * mean).clamp_min(0)
    s = torch.rsqrt(var + 1e-5) * gamma.double()
    t = beta.double() - mean * s
    a = torch.relu(A.double() * s + t)
    da = (A.double().abs() * s.abs() + t.abs()) * 2.0 ** -23
    bound = split * (a.abs() @ W.double().abs().t() + b.double().abs()) + da @ W.double().abs().t()
    return a, a @ W.double().t() + b.double(), bound


@pytest.mark.parametrize("shape", [(128, 64, 256), (300, 96, 256), (1000, 512, 768), (777, 1024, 512), (4100, 2048, 256), (33000, 160, 256),
                                   (16, 128, 32), (1000, 128, 32), (70001, 128, 32),
                                   (100, 64, 128), (1000, 512, 128), (2049, 96, 128), (50001, 512, 128),
                                   (70000, 64, 128), (100000, 64, 128)])     # tile heights 160 and 224 (staged_tile_rows)
def test_staged_matches_float64(shape):
    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N)
    A = torch.randn(M, K, device="cuda", generator=g) * (1 + 4 * torch.rand(1, K, device="cuda", generator=g)) + \
        torch.randn(1, K, device="cuda", generator=g)
    gamma = 0.5 + torch.rand(K, device="cuda", generator=g)
    beta = 0.2 * torch.randn(K, device="cuda", generator=g)
    beta[0] = -50.0                                           # a column BatchNorm + ReLU kills entirely
    beta[1] = 30.0                                            # ... and one that sets the operand scale
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    W[N // 3] *= 1e-3                                          # rows of very different magnitude: per-row weight scales
    b = torch.randn(N, device="cuda", generator=g)
    Y, st, ymax, st_in = _run(A, gamma, beta, W, b)
    # column 1 (beta = 30) dominates every dot product: with few terms nothing averages and the budget is the worst case of
    # two truncated two-piece operands (2^-20 each); from K = 512 on the statistical 5e-7 holds as for the other kernels
    a, ref, bound = _ref(A, st_in, gamma, beta, W, b, float(M), split=2.0 ** -19 if K < 512 else 5e-7)
    assert torch.isfinite(Y).all()
    assert (a[:, 0] == 0).all()
    err = ((Y.double() - ref).abs() / bound).max().item()
    assert err < 1.0, err
    assert ymax == Y.abs().max().item()
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * Y.abs().max().item() * M)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-9)


@pytest.mark.parametrize("shape", [(128, 128, 256), (1000, 512, 768), (777, 1024, 512), (4100, 2048, 256), (20011, 192, 512)])
def test_laboratory_second_form_matches_float64(shape):
    """The laboratory's second form of the role-split kernel (measured equal to the product's: profiles/r04_staged2_ab.txt):
    same bound as the product kernel, ragged tile heights, odd counts of two-k-tile intervals (K = 192: three)."""
    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + 5 * N)
    A = torch.randn(M, K, device="cuda", generator=g) * (1 + 4 * torch.rand(1, K, device="cuda", generator=g)) + \
        torch.randn(1, K, device="cuda", generator=g)
    gamma = 0.5 + torch.rand(K, device="cuda", generator=g)
    beta = 0.2 * torch.randn(K, device="cuda", generator=g)
    beta[0] = -50.0
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y, st, ymax, st_in = _run(A, gamma, beta, W, b, lab=True)
    a, ref, bound = _ref(A, st_in, gamma, beta, W, b, float(M), split=2.0 ** -19 if K < 512 else 5e-7)
    assert torch.isfinite(Y).all()
    assert ((Y.double() - ref).abs() / bound).max().item() < 1.0
    assert ymax == Y.abs().max().item()
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * Y.abs().max().item() * M)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-9)


def test_staged_strided_rows_outlier_and_foreign_statistics():
    """A with a row stride > K, one element 1e4 x the rest (it sets the scale: everything else keeps >= 22 - 14 bits ...
    judged by the bound), and statistics that are NOT A's own (a shard normalises with the whole graph's statistics)."""
    M, K, N = 515, 256, 256
    g = torch.Generator(device="cuda").manual_seed(9)
    big = torch.randn(M, K + 32, device="cuda", generator=g)
    A = big[:, :K]
    A[17, 5] = 1e4
    gamma = torch.ones(K, device="cuda")
    beta = torch.zeros(K, device="cuda")
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.zeros(N, device="cuda")
    from mtmc_mpn import _lib
    lib = _lib.load()
    other = torch.randn(4 * M, K, device="cuda", generator=g) * 2 + 0.3
    st_in = torch.cat([other.double().sum(0), (other.double() ** 2).sum(0)]).contiguous()
    count = float(4 * M)
    Y = torch.empty(M, N, device="cuda")
    work = torch.empty(4 * N * K + 4 * N + 512, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    _lib.check(lib.mtmc_linear_staged_raw(A.data_ptr(), A.stride(0), st_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), count,
                                          W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(),
                                          scr.data_ptr(), None, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    # one term dominates the outlier's row, so nothing averages: the budget is the WORST case of two truncated two-piece
    # operands, 2^-20 each (still 50 x below fp32's own rounding of a 256-term dot product relative to sum |a||w|)
    a, ref, bound = _ref(A, st_in, gamma, beta, W, b, count, split=2.0 ** -19)
    assert ((Y.double() - ref).abs() / bound.clamp_min(1e-30)).max().item() < 1.0


def test_staged_rejects_bad_shapes():
    A = torch.randn(200, 100, device="cuda")
    with pytest.raises(RuntimeError):
        _run(A, torch.ones(100, device="cuda"), torch.zeros(100, device="cuda"), torch.randn(256, 100, device="cuda"),
             torch.zeros(256, device="cuda"))                                   # K % 32 != 0
    A = torch.randn(200, 128, device="cuda")
    with pytest.raises(RuntimeError):
        _run(A, torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), torch.randn(192, 128, device="cuda"),
             torch.zeros(192, device="cuda"))                                   # N neither a multiple of 256 nor 128
