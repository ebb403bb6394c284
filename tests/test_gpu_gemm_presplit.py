"""Pre-split first-layer GEMM against float64: the product kernel (csrc/gemm_presplit.hip, mtmc_linear_presplit_raw) and
every A/B variant of the kernel laboratory (csrc/lab/presplit_lab.hip, libmtmc_lab.so: mtmc_lab_linear_presplit_raw) --
ragged edges, rows of very different magnitude, zero rows.  Same bound as the in-loop split kernel:
|err| <= 3e-7 * (sum_k |a||w| + |b|).  `variant`: "product" or a laboratory variant number (negative: reuse the planes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, A, W, b, variant, work=None):
    import lab_lib
    from mtmc_mpn import _lib
    M, K = A.shape
    N = W.shape[0]
    Y = torch.full((M, N), float("nan"), device="cuda")
    if work is None:
        work = torch.empty(4 * M * K + 4 * N * K + 4 * (M + N) + 1024, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(48, dtype=torch.int32, device="cuda")
    st = torch.empty(2 * N, dtype=torch.float64, device="cuda")
    args = (A.data_ptr(), A.stride(0), W.data_ptr(), b.data_ptr(), Y.data_ptr(), M, K, N, work.data_ptr(), work.numel(),
            scr.data_ptr(), st.data_ptr())
    if variant in ("product", "product-reuse"):
        _lib.check(lib.mtmc_linear_presplit_raw(*args, 1 if variant == "product-reuse" else 0, torch.cuda.current_stream().cuda_stream))
    else:
        rc = lab_lib.load_lab().mtmc_lab_linear_presplit_raw(*args, variant, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mtmc_lab_linear_presplit_raw: code {rc}")
    torch.cuda.synchronize()
    return Y, st, scr[32:48].view(torch.float32).max().item(), work


@pytest.mark.parametrize("shape", [(129, 64, 128), (300, 512, 130), (777, 2048, 1024), (1030, 1024, 520)])
@pytest.mark.parametrize("variant", ["product", 0, 2, 3, 4, 8, 9, 10, 11])
def test_presplit_matches_float64(shape, variant):
    from mtmc_mpn import _lib
    lib = _lib.load()
    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    A = torch.randn(M, K, device="cuda", generator=g) * torch.exp(3 * torch.randn(M, 1, device="cuda", generator=g))
    A[M // 2] = 0                                            # an all-zero row: scale 1, exact zeros
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y, st, ymax, _ = _run(lib, A, W, b, variant)
    ref = A.double() @ W.double().t() + b.double()
    bound = A.double().abs() @ W.double().abs().t() + b.double().abs()
    assert torch.isfinite(Y).all()
    assert ((Y.double() - ref).abs() / bound).max().item() < 3e-7
    assert torch.equal(Y[M // 2], b)                         # zero row: bias only, exactly
    assert ymax == Y.abs().max().item()
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * Y.abs().max().item() * M)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-9)


@pytest.mark.parametrize("M", [4_100, 20_000, 45_000, 70_001, 100_000])
def test_presplit_tile_heights(M):
    """The product kernel picks its tile height per launch (presplit_tile_rows: 144 .. 256 rows, the one that fills the last
    round of workgroups): 144, 160, 240, 224 and 224 rows for these row counts at four column tiles, ragged last tiles
    included."""
    from mtmc_mpn import _lib
    lib = _lib.load()
    K, N = 64, 1024
    g = torch.Generator(device="cuda").manual_seed(M)
    A = torch.randn(M, K, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y, st, ymax, _ = _run(lib, A, W, b, "product")
    ref = A.double() @ W.double().t() + b.double()
    bound = A.double().abs() @ W.double().abs().t() + b.double().abs()
    assert torch.isfinite(Y).all()
    # 64 terms do not average: the budget is the worst case of two truncated two-piece operands (2^-20 each), as for the
    # short-K cases of tests/test_gpu_gemm_staged.py; the statistical 3e-7 of the other tests holds from K = 512 on
    assert ((Y.double() - ref).abs() / bound).max().item() < 2.0 ** -19
    assert ymax == Y.abs().max().item()
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * Y.abs().max().item() * M)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-9)


@pytest.mark.parametrize("M", [20000, 4100])
def test_presplit_tile_heights_long_k_tight_bound(M):
    """The variable-height paths (row blocks skipped per wave, skipped second DMA pass, clamped row offsets of the last
    tile) at a K where the statistical 3e-7 bound of the other tests applies -- a wrong-row or stale-LDS error in the short
    tiles cannot hide inside the relaxed short-K budget above.  20000 rows at two column tiles pick 160-row tiles
    (125 x 2 = 250 tiles: one round), 4100 rows 144-row tiles with a ragged last one."""
    from mtmc_mpn import _lib
    lib = _lib.load()
    K, N = 512, 512
    g = torch.Generator(device="cuda").manual_seed(M + 1)
    A = torch.randn(M, K, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y, st, ymax, _ = _run(lib, A, W, b, "product")
    ref = A.double() @ W.double().t() + b.double()
    bound = A.double().abs() @ W.double().abs().t() + b.double().abs()
    assert torch.isfinite(Y).all()
    assert ((Y.double() - ref).abs() / bound).max().item() < 3e-7
    assert ymax == Y.abs().max().item()
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-9, atol=1e-9 * Y.abs().max().item() * M)


def test_presplit_strided_rows_and_reused_planes():
    from mtmc_mpn import _lib
    lib = _lib.load()
    M, K, N = 400, 256, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(M, K + 64, device="cuda", generator=g)
    A = big[:, :K]                                           # row stride K + 64
    W = torch.randn(N, K, device="cuda", generator=g)
    b = torch.zeros(N, device="cuda")
    Y0, _, _, work = _run(lib, A, W, b, "product")
    Y1, _, _, _ = _run(lib, A, W, b, "product-reuse", work)  # planes already in `work`
    assert torch.equal(Y0, Y1)
    Y2, _, _, _ = _run(lib, A, W, b, -1, work)               # laboratory variant 0 on the same planes
    assert (Y2 - Y0).abs().max().item() <= 1e-5 * Y0.abs().max().item()
    ref = A.double() @ W.double().t()
    assert ((Y0.double() - ref).abs() / (A.double().abs() @ W.double().abs().t())).max().item() < 3e-7


def test_presplit_rejects_bad_shapes():
    from mtmc_mpn import _lib
    lib = _lib.load()
    A = torch.randn(8, 96, device="cuda")
    W = torch.randn(8, 96, device="cuda")
    b = torch.zeros(8, device="cuda")
    with pytest.raises(RuntimeError):
        _run(lib, A, W, b, "product")                        # K % 64 != 0
    A = torch.randn(8, 64, device="cuda")
    W = torch.randn(8, 64, device="cuda")
    with pytest.raises(RuntimeError):
        _run(lib, A, W, b, "product", work=torch.empty(64, dtype=torch.uint8, device="cuda"))   # work too small


@pytest.mark.parametrize("M,K", [(77, 64), (1030, 512), (4097, 2048)])
def test_plane_layout_and_split_are_as_documented(M, K):
    """The fp16 planes in `work` follow the layout csrc/gemm_presplit.hip documents (kPlaneKT note): the eight halves
    k..k+7 of row r sit at ((k/32)*rows + r)*32 + 8*(((k%32)/8) ^ swz(r)), swz(r) = {0, 2, 3, 1}[(r>>2)&3] (round 5: the
    permutation that makes the 16 x 16 x 32 fragment read conflict-free, lds_dma.h); h1 + h2 reproduce x * 2^(14-e) to 22 bits
    and inv[r] = 2^(e-14) with 2^(e-1) <= max|row| < 2^e."""
    from mtmc_mpn import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    A = (torch.randn(M, K, generator=g) * torch.logspace(-3, 3, M).unsqueeze(1)).cuda()
    A[M // 2] = 0.0                                              # a zero row: scale stays finite
    W = torch.randn(40, K, generator=g).cuda()
    _, _, _, work = _run(lib, A, W, torch.zeros(40, device="cuda"), "product")
    planes = work[:M * K * 4].view(torch.float16).view(2, K // 32, M, 4, 8)      # [piece][k-tile][row][slot][8]
    inv = work[M * K * 4:M * K * 4 + M * 4].view(torch.float32)
    r = torch.arange(M, device="cuda")
    swz = torch.tensor([0, 2, 3, 1], device="cuda")[(r >> 2) & 3]
    slot = torch.arange(4, device="cuda").unsqueeze(0) ^ swz.unsqueeze(1)                     # [row][logical slot] -> stored slot
    idx = slot.view(1, 1, M, 4, 1).expand(2, K // 32, M, 4, 8)
    logical = torch.gather(planes, 3, idx)                       # [piece][k-tile][row][logical slot][8]
    h = logical.permute(0, 2, 1, 3, 4).reshape(2, M, K).double()
    rec = (h[0] + h[1]) * inv.double().unsqueeze(1)
    amax = A.abs().amax(dim=1).double()
    assert (rec - A.double()).abs().max().item() <= 0.0 + (amax * 2.0 ** -21).max().item()
    assert ((rec - A.double()).abs() <= amax.unsqueeze(1) * 2.0 ** -21 + 1e-300).all()
    live = amax > 0
    e = torch.log2(inv.double()[live]) + 14
    assert (e == e.round()).all()                                # powers of two
    assert ((amax[live] < 2.0 ** e) & (amax[live] >= 2.0 ** (e - 1))).all()
