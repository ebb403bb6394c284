// Node encoder of FEW-ROW graphs (the graph the headline metric is quoted on: 450 tracklets; reference models/mlp.py:14-27
// via models/mpn.py:131, timed call inference.py:468-471) -- one launch per layer, no split along K across workgroups and
// therefore no slabs and no combine launch: 4 launches where rounds 1-4 had 7.
//
// Rounds 1-4 ran these layers on the many-tile kernel shape (64 x 64 x 64 tiles, operands split inside the k-loop, K cut
// four ways across workgroups, slabs summed by combine_stats_kernel): at 450 rows every 64-row tile re-fetched its W0
// K-slice through the fabric (71 MB for 12 MB of compulsory bytes) and each layer cost two grid-wide dependencies.
//
//   few_l0_kernel     layer 0: x . W0^T on PRE-SPLIT operands -- the planes of x are made by passenger workgroups of
//                     prep_kernel (split_body.h), the planes of W0 come from the content-verified weight-plane cache.  A
//                     workgroup owns 64 rows x 32 columns over ALL of K; both operands arrive by LDS-DMA in a five-stage
//                     ring (four k-steps = 96 KB in flight per CU), so the k-loop is paced by what a CU can take in, not by
//                     a chain of dependent round trips -- the reason the old shape cut K across workgroups.  Column tiles
//                     are bound to XCDs in groups (blockIdx % 8): an XCD's L2 keeps its 1 MB of W0 and streams x.
//   few_wave_kernel   layers >= 1: Y = relu(bn(Y_prev)) . W^T + b.  A workgroup owns 16 rows x 16 or 32 columns; its waves
//                     CUT K BETWEEN THEM (wave q takes k-steps [q NKB, (q + 1) NKB)), each loads its A rows as fp32 and its
//                     W fragments straight from the cached planes into registers (the k-tile-major, pre-swizzled plane
//                     layout makes a fragment one 16-byte load per lane), applies the input BatchNorm + ReLU, takes the
//                     wave's own |.|max for an exact power-of-two scale, splits and multiplies -- ALL loads of a wave are
//                     in flight at once and there is no barrier before the one that sums the waves' partial tiles in LDS
//                     (fixed order: reproducible).  No |Y|max hand-off between layers any more.
// Both take the fp64 column statistics of their raw output in the epilogue (atomics, as before).
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "lds_dma.h"

namespace mtmc {

#ifndef FEW_STAMP
#define FEW_STAMP 0       // 1: two workgroups record s_memtime along the kernels, per wave (tools/few_stamps.py; tools/build_variant.sh)
#endif
#if FEW_STAMP
constexpr int kFsSteps = 40, kFsL0 = 4 + 4 * kFsSteps, kFsWave = 8;
__device__ unsigned long long g_fs_l0[2 * 8 * kFsL0];        // [block slot][wave][entry, prologue issued, loop end, end | step x 4]
__device__ unsigned long long g_fs_wave[2 * 8 * kFsWave];    // [block slot][wave][point]
#define FS_L0(pt)                                                                                             \
  do { if (fs_slot >= 0 && lane == 0) g_fs_l0[(fs_slot * 8 + wid) * kFsL0 + (pt)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FS_L0_STEP(kt, pt)                                                                                    \
  do { if (fs_slot >= 0 && lane == 0 && (kt) < kFsSteps) g_fs_l0[(fs_slot * 8 + wid) * kFsL0 + 4 + 4 * (kt) + (pt)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FS_WAVE(pt)                                                                                           \
  do { if (fs_slot >= 0 && lane == 0 && wid < 8) g_fs_wave[(fs_slot * 8 + wid) * kFsWave + (pt)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FS_L0(pt) do {} while (0)
#define FS_L0_STEP(kt, pt) do {} while (0)
#define FS_WAVE(pt) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ------------------------------------------------------------------------------------------------
// layer 0
// ------------------------------------------------------------------------------------------------
// 512 threads split by ROLE (a workgroup's waves go to the SIMDs cyclically: every SIMD hosts one wave of each role):
//   waves 0-3, consumers: wave w multiplies rows 16 w .. + 15 of the 64 x 32 tile (2 accumulators): per k-step of 64, twelve
//                         ds_read_b128 fragment reads and twelve v_mfma_f32_16x16x32_f16;
//   waves 4-7, loaders:   six LDS-DMA instructions (1 KB each) per wave and k-step, FOUR k-steps ahead, and the vmcnt waits.
// One barrier per k-step.  Why the split (profiles/r05_few_stamps_v1.txt: the first form had all four waves do everything, in
// turn): an LDS-DMA instruction costs the issuing wave ~64 cycles and the CU's address path takes the workgroup's 24 per
// k-step one after the other (385 cycles), during which no wave multiplied, and while the waves multiplied (510 cycles) or sat
// in their vmcnt wait and the barrier (340) nothing was issued -- the CU took in 24 KB per 1230 cycles, 20 B / clk, whatever
// the tile shape (64 x 64 tiles on half the CUs: 32 KB per 1370 cycles).  Loaders that do nothing else keep the address path
// busy while the consumers multiply.
constexpr int kL0BM = 64, kL0BN = 32, kL0Stages = 5;
constexpr int kL0AImg = kL0BM * 64, kL0WImg = kL0BN * 64;             // bytes of one [rows][32 halves] image
constexpr int kL0Stage = 4 * kL0AImg + 4 * kL0WImg;                   // 2 planes x 2 k-tiles of A and of W: 24 KB
constexpr int kL0Ops = 6;                                             // LDS-DMA instructions per loader wave and k-step

__global__ __launch_bounds__(512) void few_l0_kernel(FewL0Params p, int tiles_m, int tiles_n, int cpx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // column tiles in groups of cpx per XCD (blocks b, b + 8, ... share an XCD under round-robin placement: speed only)
  int tm, tn;
  if (cpx == -1) {                     // (experiment, MTMC_FEW_L0_MAP=1: 4 row tiles x 8 column tiles per XCD; 8 x 32 tiles only)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tm = 4 * (xcd & 1) + (slot >> 3); tn = 8 * (xcd >> 1) + (slot & 7);
  } else if (cpx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tn = xcd * cpx + slot % cpx; tm = slot / cpx;
  } else {
    tn = blockIdx.x % tiles_n; tm = blockIdx.x / tiles_n;
  }
  if (tn >= tiles_n || tm >= tiles_m) return;
  const int64_t m0 = (int64_t)tm * kL0BM;
  const int n0 = tn * kL0BN;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wid >= 4;
  const int wm = wid & 3;                                                        // row block (consumers) / share of the DMA (loaders)
  const int r16 = lane & 15, ks = lane >> 4;
#if FEW_STAMP
  const int fs_slot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 + 3 ? 1 : -1);
#endif
  FS_L0(0);
  const int nk = p.K / 64;

  // LDS-DMA sources.  Loader l brings in rows 16 l .. + 15 of all four images (plane, k-tile) of A, and both 16-row pieces of
  // image (plane l / 2, k-tile l % 2) of W.
  const int64_t a_rows_left = p.M - 1 - m0;                                      // rows past M: the last valid row again
  const int a_row = wm * 16 + (lane >> 2);
  const unsigned off_a = (unsigned)((a_row < a_rows_left ? a_row : a_rows_left) * 64 + (lane & 3) * 16);
  const unsigned off_w0 = (unsigned)((lane >> 2) * 64 + (lane & 3) * 16), off_w1 = off_w0 + 16 * 64;
  const int64_t a_plane = p.M * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;    // bytes
  const int64_t a_kt = p.M * 64, w_kt = (int64_t)p.Nout * 64;                             // bytes per k-tile of 32
  const char* a_base = reinterpret_cast<const char*>(p.Ah) + m0 * 64;
  const char* w_base = reinterpret_cast<const char*>(p.Wh) + (int64_t)n0 * 64 + (wm >> 1) * w_plane;
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto issue = [&](int kt, int stage) {
    const unsigned st = lds0 + stage * kL0Stage;
#pragma unroll
    for (int img = 0; img < 4; ++img)                                            // img = plane * 2 + k-tile
      lds_dma16(a_base + (img >> 1) * a_plane + (int64_t)(2 * kt + (img & 1)) * a_kt, off_a, st + img * kL0AImg + wm * 1024);
    const char* wb = w_base + (int64_t)(2 * kt + (wm & 1)) * w_kt;
    lds_dma16(wb, off_w0, st + 4 * kL0AImg + wm * kL0WImg);
    lds_dma16(wb, off_w1, st + 4 * kL0AImg + wm * kL0WImg + 1024);
  };
  // the loader's share of k-step kt has landed once at most the (younger) steps kt + 1 .. kt + Stages - 2 are outstanding
  auto wait_step = [&](int kt) {
    const int younger = nk - 1 - kt < kL0Stages - 2 ? nk - 1 - kt : kL0Stages - 2;
    if (younger >= 3) wait_vmcnt<3 * kL0Ops>();
    else if (younger == 2) wait_vmcnt<2 * kL0Ops>();
    else if (younger == 1) wait_vmcnt<1 * kL0Ops>();
    else wait_vmcnt<0>();
  };

  const int so = (ks ^ plane_swz(r16)) * 16;                                     // the stored swizzle (lds_dma.h)
  const int a_frag = (wm * 16 + r16) * 64 + so, b_frag = 4 * kL0AImg + r16 * 64 + so;
  f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
  auto multiply = [&](int stage) {
    const unsigned char* st = smem + stage * kL0Stage;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 a[2], b[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) a[q] = *reinterpret_cast<const f16x8*>(st + (q * 2 + kk) * kL0AImg + a_frag);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(st + b_frag + (q * 2 + kk) * kL0WImg + j * 16 * 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[j][0], acc[j], 0, 0, 0);   // smallest terms first
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][0], acc[j], 0, 0, 0);
      }
    }
  };

  // what the epilogue needs from memory, requested by the consumers before anything else
  float ia[4] = {0.f, 0.f, 0.f, 0.f}, iw[2] = {0.f, 0.f}, bias[2] = {0.f, 0.f};
  if (loader) {
    for (int s = 0; s < kL0Stages - 1 && s < nk; ++s) issue(s, s);
    FS_L0(1);
    wait_step(0);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + wm * 16 + 4 * ks + r;
      ia[r] = p.inv_a[row < p.M ? row : p.M - 1];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      iw[j] = p.inv_w[n0 + j * 16 + r16];
      bias[j] = p.bias[n0 + j * 16 + r16];
    }
    FS_L0(1);
  }
  __syncthreads();                         // k-step 0 is in LDS
  for (int kt = 0; kt < nk; ++kt) {
    FS_L0_STEP(kt, 0);
    if (loader) {
      // stage (kt - 1) % Stages was read in the interval the barrier above closed: refill it, then see step kt + 1 land
      if (kt + kL0Stages - 1 < nk) issue(kt + kL0Stages - 1, (kt + kL0Stages - 1) % kL0Stages);
      FS_L0_STEP(kt, 1);
      if (kt + 1 < nk) wait_step(kt + 1);
    } else {
      multiply(kt % kL0Stages);
      FS_L0_STEP(kt, 1);
    }
    FS_L0_STEP(kt, 2);
    __syncthreads();
    FS_L0_STEP(kt, 3);
  }
  FS_L0(2);

  // ---- epilogue (consumers): undo the row scales, + bias, raw Y, fp64 column statistics
  double* colred = reinterpret_cast<double*>(smem);                              // [4 waves][2][32]  (every stage has been read)
  if (!loader) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + j * 16 + r16;
      double cs = 0, cq = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 16 + 4 * ks + r;
        if (row < p.M) {
          const float y = fmaf(acc[j][r] * ia[r], iw[j], bias[j]);
          p.Y[row * p.ldy + col] = y;
          cs += y;
          cq += (double)y * y;
        }
      }
      cs += __shfl_xor(cs, 16, 64); cq += __shfl_xor(cq, 16, 64);
      cs += __shfl_xor(cs, 32, 64); cq += __shfl_xor(cq, 32, 64);
      if (lane < 16) {
        colred[(wm * 2 + 0) * kL0BN + j * 16 + r16] = cs;
        colred[(wm * 2 + 1) * kL0BN + j * 16 + r16] = cq;
      }
    }
  }
  __syncthreads();
  if (p.stats_out != nullptr && threadIdx.x < 2 * kL0BN) {
    const int which = threadIdx.x / kL0BN, cl = threadIdx.x % kL0BN;
    double s = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += colred[(w * 2 + which) * kL0BN + cl];
    unsafeAtomicAdd(p.stats_out + which * p.Nout + n0 + cl, s);
  }
  FS_L0(3);
}

bool few_l0_shape(int K, int Nout) { return K % 64 == 0 && K >= 64 && K <= 2048 && Nout % kL0BN == 0 && Nout >= kL0BN; }

int launch_few_l0(const FewL0Params& p, hipStream_t s) {
  if (p.M < 1 || !few_l0_shape(p.K, p.Nout) || !p.Ah || !p.Wh || !p.inv_a || !p.inv_w || !p.bias || !p.Y) return 1;
  const int tiles_m = (int)((p.M + kL0BM - 1) / kL0BM), tiles_n = p.Nout / kL0BN;
  int cpx = tiles_n >= 8 ? (tiles_n + 7) / 8 : 0;
  int grid = cpx > 0 ? 8 * cpx * tiles_m : tiles_n * tiles_m;
  const int map = knobs().few_l0_map;          // A/B of the tile -> XCD binding
  if (map == 1 && tiles_m == 8 && tiles_n == 32) cpx = -1;
  if (map == 2) { cpx = 0; grid = tiles_n * tiles_m; }
  const size_t lds = (size_t)kL0Stages * kL0Stage;
  if (!allow_big_lds(reinterpret_cast<const void*>(few_l0_kernel), (int)lds)) return MTMC_E_HIP;
  hipLaunchKernelGGL(few_l0_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n, cpx);
  return MTMC_OK;
}

// ------------------------------------------------------------------------------------------------
// layers >= 1
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_f32(float v) {   // operands >= 0: lanes without a source read 0 (bound_ctrl)
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true)));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = dpp_max_f32<0x111, 0xf>(v);   // row_shr:1
  v = dpp_max_f32<0x112, 0xf>(v);   // row_shr:2
  v = dpp_max_f32<0x114, 0xf>(v);   // row_shr:4
  v = dpp_max_f32<0x118, 0xf>(v);   // row_shr:8
  v = dpp_max_f32<0x142, 0xa>(v);   // row_bcast:15
  v = dpp_max_f32<0x143, 0xc>(v);   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// DROP: the encoder's Dropout on the A operand compiled in (training; counter-based, as in gemm_bn_f16x3_kernel)
template <int NKB, int CB, int RB, bool DROP>     // k-steps of 32 per wave; 16-column blocks and 16-row blocks per workgroup
__global__ __launch_bounds__(512) void few_wave_kernel(FewWaveParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x >= p.tiles) {                       // passenger workgroups (256 threads): the edge encoder's enc2 job
    enc2_body(p.pass_enc, p.pass_attr, p.pass_edges, p.pass_e_total, p.pass_stat, (int)blockIdx.x - p.tiles, p.pass_blocks);
    return;
  }
  if (DROP) drop_resolve(p.drop_in);
  constexpr int KW = 32 * NKB;                            // K columns of one wave
  const int nw = (int)blockDim.x >> 6;
  float* aff = reinterpret_cast<float*>(smem);            // [nw][2][KW]
  f32x4v* red = reinterpret_cast<f32x4v*>(smem + (size_t)nw * 2 * KW * sizeof(float));   // [nw][RB][CB][64]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int tiles_n = p.Nout / (16 * CB);
  const int tn = blockIdx.x % tiles_n, rt = blockIdx.x / tiles_n;   // (a column tile stays on XCD tn % 8 when tiles_n % 8 == 0)
  const int64_t row0 = (int64_t)rt * 16 * RB;
  const int n0 = tn * 16 * CB;
  const int k0 = wid * KW;
#if FEW_STAMP
  const int fs_slot = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == p.tiles / 2 + 3 ? 1 : -1);
#endif
  FS_WAVE(0);

  // ---- every load of this wave, up front.  First what the epilogue needs (it would otherwise queue behind everything else)
  // and the input statistics, then the A rows (fp32) and the W fragments (fp16 planes).
  float e_iw[RB * CB], e_bias[RB * CB];                   // (the epilogue's jobs of this wave: block jb = wid + nw * t)
#pragma unroll
  for (int t = 0; t < RB * CB; ++t) {
    const int jb = wid + nw * t, col = n0 + 16 * (jb % CB) + i16;
    e_iw[t] = jb < RB * CB ? p.inv_w[col] : 0.f;
    e_bias[t] = jb < RB * CB ? p.bias[col] : 0.f;
  }
  float* my_aff = aff + wid * 2 * KW;
#pragma unroll
  for (int i = 0; i < (KW + 63) / 64; ++i) {
    const int kk = lane + 64 * i;
    if (kk < KW) {
      const int k = k0 + kk;
      float sv, tv;
      bn_affine(p.stats_in[k], p.stats_in[p.K + k], p.count, p.gamma_in[k], p.beta_in[k], sv, tv);
      my_aff[kk] = sv;
      my_aff[KW + kk] = tv;
    }
  }
  float4 av[RB][NKB][2];
  int64_t arows[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t arow = row0 + 16 * rb + i16 < p.M ? row0 + 16 * rb + i16 : p.M - 1;
    arows[rb] = arow;
    const float* asrc = p.A + arow * p.lda + k0 + 8 * g;
#pragma unroll
    for (int s = 0; s < NKB; ++s) {
      av[rb][s][0] = *reinterpret_cast<const float4*>(asrc + 32 * s);
      av[rb][s][1] = *reinterpret_cast<const float4*>(asrc + 32 * s + 4);
    }
  }
  f16x8 bw[NKB][CB][2];
  {
    const int64_t plane = (int64_t)p.Nout * p.K;                                 // halves
    const int sw = (g ^ plane_swz(i16)) * 8;
#pragma unroll
    for (int s = 0; s < NKB; ++s)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const _Float16* src = p.Wh + ((int64_t)(wid * NKB + s) * p.Nout + n0 + 16 * j + i16) * kPlaneKT + sw;
        bw[s][j][0] = *reinterpret_cast<const f16x8*>(src);
        bw[s][j][1] = *reinterpret_cast<const f16x8*>(src + plane);
      }
  }
  FS_WAVE(1);                                                   // statistics landed, affine in LDS
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // (a wave's LDS accesses execute in order; this pins the compiler)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    // ---- input BatchNorm + ReLU in registers, the wave's own |.|max of the block -> an exact power-of-two scale
    float xv[NKB][8];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < NKB; ++s) {
      const float4 s0 = *reinterpret_cast<const float4*>(my_aff + 32 * s + 8 * g), s1 = *reinterpret_cast<const float4*>(my_aff + 32 * s + 8 * g + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(my_aff + KW + 32 * s + 8 * g), t1 = *reinterpret_cast<const float4*>(my_aff + KW + 32 * s + 8 * g + 4);
      const float x[8] = {av[rb][s][0].x, av[rb][s][0].y, av[rb][s][0].z, av[rb][s][0].w, av[rb][s][1].x, av[rb][s][1].y, av[rb][s][1].z, av[rb][s][1].w};
      const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[s][j] = fmaxf(fmaf(x[j], sv[j], tv[j]), 0.f);
      if (DROP) {                                              // eight consecutive elements of a row: two hashes
        const unsigned long long idx = (unsigned long long)arows[rb] * p.K + k0 + 32 * s + 8 * g;
        float lo[4] = {xv[s][0], xv[s][1], xv[s][2], xv[s][3]}, hi[4] = {xv[s][4], xv[s][5], xv[s][6], xv[s][7]};
        drop_apply4(p.drop_in, p.drop_stream, idx, lo);
        drop_apply4(p.drop_in, p.drop_stream, idx + 4, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[s][j] = lo[j]; xv[s][4 + j] = hi[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, xv[s][j]);
    }
    m = wave_max_nonneg(m);
    if (rb == 0) FS_WAVE(2);                                    // A rows landed, activated, |.|max known
    int ea = 0;
    if (m > 0.f && m < 3e38f) ea = __builtin_amdgcn_frexp_expf(m);
    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
    const float sa = ldexpf(1.f, 14 - ea), inv_sa = ldexpf(1.f, ea - 14);

    f32x4v acc[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKB; ++s) {
      f16x8 a1, a2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xa = xv[s][2 * j] * sa, xb = xv[s][2 * j + 1] * sa;
        const h2_t hi = __builtin_amdgcn_cvt_pkrtz(xa, xb);
        const h2_t lo = __builtin_amdgcn_cvt_pkrtz(xa - (float)hi[0], xb - (float)hi[1]);
        a1[2 * j] = (_Float16)hi[0]; a1[2 * j + 1] = (_Float16)hi[1];
        a2[2 * j] = (_Float16)lo[0]; a2[2 * j + 1] = (_Float16)lo[1];
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bw[s][j][0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bw[s][j][1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bw[s][j][0], acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      f32x4v v = acc[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= inv_sa;
      red[((wid * RB + rb) * CB + j) * 64 + lane] = v;
    }
  }
  FS_WAVE(3);                                                   // W fragments landed, MFMAs done, partial tiles in LDS
  __syncthreads();
  FS_WAVE(4);

  // ---- the waves' partial tiles added in a fixed order, block jb = (row block, column block) by wave jb % nw: bias, raw Y,
  // column statistics
#pragma unroll
  for (int t = 0; t < RB * CB; ++t) {
    const int jb = wid + nw * t;
    if (jb < RB * CB) {
      const int rb = jb / CB, cb = jb % CB;
      const int col = n0 + 16 * cb + i16;
      f32x4v y = red[jb * 64 + lane];
      for (int q = 1; q < nw; ++q) {
        const f32x4v v = red[(q * RB * CB + jb) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] += v[r];
      }
      double cs = 0, cq = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 16 * rb + 4 * g + r;        // a lane holds rows 4 g + r of column i16
        if (row < p.M) {
          const float yy = fmaf(y[r], e_iw[t], e_bias[t]);
          p.Y[row * p.ldy + col] = yy;
          cs += yy;
          cq += (double)yy * yy;
        }
      }
      cs += __shfl_xor(cs, 16, 64); cq += __shfl_xor(cq, 16, 64);
      cs += __shfl_xor(cs, 32, 64); cq += __shfl_xor(cq, 32, 64);
      if (lane < 16 && p.stats_out) {
        unsafeAtomicAdd(p.stats_out + col, cs);
        unsafeAtomicAdd(p.stats_out + p.Nout + col, cq);
      }
    }
  }
  FS_WAVE(5);
}

// k-steps per wave for a layer of depth K: at most 8 waves per workgroup (512 threads: the register budget of the <4, 2, 2> form)
int few_wave_nkb(int K) {
  if (K % 128 == 0 && K / 128 >= 2 && K / 128 <= 8) return 4;
  if (K % 32 == 0 && K / 32 <= 8) return 1;
  if (K % 256 == 0 && K / 256 <= 8) return 8;
  return 0;
}
bool few_wave_shape(int K, int Nout) { return K >= 32 && few_wave_nkb(K) > 0 && Nout % 16 == 0 && Nout >= 16; }

// 16-column blocks per workgroup: two where that still leaves at least a workgroup per CU (or the width asks for it)
static int few_wave_cb(int64_t M, int K, int Nout) {
  if (few_wave_nkb(K) == 8 || Nout % 32 != 0) return 1;
  const int64_t row_blocks = (M + 15) / 16;
  return row_blocks * (Nout / 32) >= 256 ? 2 : 1;
}
// 16-row blocks per workgroup: two (the W fragments stay in registers for both: half the W traffic, which is what bounds the
// kernel when there are more workgroups than the chip holds at once) where that still leaves nearly a workgroup per CU
static int few_wave_rb(int64_t M, int K, int Nout, int cb) {
  const int forced = knobs().few_wave_rb;
  if (few_wave_nkb(K) != 4 || cb != 2) return 1;
  if (forced == 1 || forced == 2) return forced;
  const int64_t row_blocks = (M + 15) / 16;
  return row_blocks * (Nout / 32) >= 400 ? 2 : 1;
}
int few_wave_threads(int K) { const int nkb = few_wave_nkb(K); return nkb > 0 ? 64 * (K / (32 * nkb)) : 0; }

int launch_few_wave(const FewWaveParams& p0, hipStream_t s) {
  FewWaveParams p = p0;
  if (p.M < 1 || !few_wave_shape(p.K, p.Nout) || !p.A || !p.Wh || !p.inv_w || !p.bias || !p.Y || !p.stats_in || !p.gamma_in ||
      !p.beta_in || (p.lda & 3) || ((uintptr_t)p.A & 15))
    return 1;
  const int nkb = few_wave_nkb(p.K), cb = few_wave_cb(p.M, p.K, p.Nout), rb = few_wave_rb(p.M, p.K, p.Nout, cb);
  const int nw = p.K / (32 * nkb);
  p.tiles = (int)((p.M + 16 * rb - 1) / (16 * rb)) * (p.Nout / (16 * cb));
  const bool carries = p.pass_blocks > 0 && nw == 4;     // enc2_body is written for 256-thread workgroups
  const bool job_behind = p.pass_blocks > 0 && !carries;
  if (!carries) p.pass_blocks = 0;
  const size_t lds = (size_t)nw * 2 * 32 * nkb * sizeof(float) + (size_t)nw * rb * cb * 64 * sizeof(f32x4v);
  const dim3 grid(p.tiles + p.pass_blocks), block(64 * nw);
  const bool drop = p.drop_in.on != 0;
#define FEW_WAVE_LAUNCH(A, B, C)                                                                        \
  do {                                                                                                  \
    if (drop) hipLaunchKernelGGL((few_wave_kernel<A, B, C, true>), grid, block, lds, s, p);             \
    else hipLaunchKernelGGL((few_wave_kernel<A, B, C, false>), grid, block, lds, s, p);                 \
  } while (0)
  if (nkb == 4 && cb == 2 && rb == 2) FEW_WAVE_LAUNCH(4, 2, 2);
  else if (nkb == 4 && cb == 2) FEW_WAVE_LAUNCH(4, 2, 1);
  else if (nkb == 4) FEW_WAVE_LAUNCH(4, 1, 1);
  else if (nkb == 1 && cb == 2) FEW_WAVE_LAUNCH(1, 2, 1);
  else if (nkb == 1) FEW_WAVE_LAUNCH(1, 1, 1);
  else FEW_WAVE_LAUNCH(8, 1, 1);
#undef FEW_WAVE_LAUNCH
  if (job_behind) launch_enc2(p0.pass_enc, p0.pass_attr, p0.pass_edges, p0.pass_e_total, p0.pass_stat, s);
  return MTMC_OK;
}

// Few-row graphs in eval mode with a weight-plane cache: every layer on the kernels above (all or nothing: they hand no |Y|max
// to the in-loop kernel).  kFewRowsMax: measured crossover against the split-K shape.
bool few_rows_path(int64_t rows, int n_layers, const int* in_dim, const int* out_dim) {
  const Knobs& kn = knobs();
  if (kn.gemm_no_few || kn.gemm_fp32 || kn.gemm_no_f16) return false;
  if (rows < 2 || rows > kn.few_rows_max) return false;
  if (!few_l0_shape(in_dim[0], out_dim[0])) return false;
  for (int l = 1; l < n_layers; ++l)
    if (!few_wave_shape(in_dim[l], out_dim[l]) || in_dim[l] > 2048) return false;
  return true;
}

}  // namespace mtmc

#if FEW_STAMP
extern "C" int mtmc_dbg_few_stamps(unsigned long long* l0, unsigned long long* wave) {   // host buffers of 2*8*164 and 2*8*8 entries
  if (hipMemcpyFromSymbol(l0, HIP_SYMBOL(mtmc::g_fs_l0), sizeof(mtmc::g_fs_l0)) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(wave, HIP_SYMBOL(mtmc::g_fs_wave), sizeof(mtmc::g_fs_wave));
}
#endif
