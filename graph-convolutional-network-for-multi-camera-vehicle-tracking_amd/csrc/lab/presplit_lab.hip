// Kernel laboratory of the pre-split first-layer GEMM: the variants gemm_f16p_m16_kernel (../gemm_presplit.hip, the one the
// forward runs) was chosen against, and timing experiments whose results are deliberately WRONG.  Built into
// libmtmc_lab.so -- never into libmtmc_mpn.so -- for tools/presplit_time.py and tests/test_gpu_gemm_presplit.py; DESIGN.md
// Appendix A quotes the measurements.  Entry point: mtmc_lab_linear_presplit_raw(..., variant, stream):
//   0 256x256 two-stage loop (32x32x16 MFMA) . 2 / 3 / 4 / 8 other tile / stage counts . 9 mid-tile barrier . 10 ping-pong .
//   11 the product kernel . 12-15 timing experiments (no MFMAs / no LDS-DMA / no fragment reads / MFMAs only), WRONG results .
//   17 right results + per-phase shader-clock averages written over Y[0][0..31].  variant < 0: reuse the planes in `work`.
#include <hip/hip_runtime.h>

#include "../../../include/mtmc_mpn.h"
#include "../common.h"
#include "../kernels.h"
#include "../lds_dma.h"

namespace mtmc {

// ------------------------------------------------------------------------------------------------
// BT x BT tile (128: 4 waves as 2x2, 64x64 each; 256: 8 waves as 2x4, 128x64 each), BK columns per k-tile, NBUF stages.
// A stage holds four images [BT rows][BK halves]: A piece 1, A piece 2, W piece 1, W piece 2.  An image is filled by
// LDS-DMA in lane order (16-byte chunk c of the image lands at byte 16*c), so the bank swizzle is applied to the
// SOURCE address: chunk `sp` of row r holds the row's 16-byte slot sp ^ g(r); the fragment reads apply the same XOR.
//   BK = 64 (128-byte rows, two rows per 256-byte bank row):  g(r) = (r >> 1) & 7
//   BK = 32 ( 64-byte rows, four rows per bank row):          g(r) = plane_swz(r) (lds_dma.h; rounds 2-4: (r >> 2) & 3)
//   BK = 16 ( 32-byte rows, eight rows per bank row):         g(r) = (r >> 3) & 1
// With these every 16-lane group of a ds_read_b128 (MI355X_MICROARCH.md, LDS) touches 16 distinct slots.
// ------------------------------------------------------------------------------------------------
// DIAG (timing experiments, results are wrong except for 7): 2 = no MFMAs (DMA + fragment reads + barriers),
// 3 = no DMA inside the loop (fragment reads + MFMAs + barriers), 4 = no fragment reads (DMA + MFMAs + barriers),
// 5 = MFMAs and barriers only, 7 = the product loop with s_memtime around its phases
template <int BT, int BK, int NBUF, int MINB, int DIAG = 0>
__global__ __launch_bounds__(BT * 2, MINB) void gemm_f16p_kernel(SplitGemmParams p, int tiles_m, int tiles_n) {
  constexpr int NT = BT * 2;                     // threads
  constexpr int WN = BT / 64;                    // waves across the tile's columns (2 rows of waves)
  constexpr int TI = BT / 64;                    // 32-row blocks per wave (rows), 2 column blocks per wave
  constexpr int SLOTS = BK / 8;                  // 16-byte slots per image row
  constexpr int ROWB = BK * 2;                   // bytes per image row
  constexpr int IMG = BT * ROWB;                 // bytes per image
  constexpr int STAGE = 4 * IMG;
  constexpr int RPI = NT / SLOTS;                // image rows one whole-block instruction covers
  constexpr int IPI = BT / RPI;                  // instructions per image
  static_assert(RPI % 16 == 0 && IPI >= 1, "swizzle period");
  static_assert(BK == kPlaneKT, "a k-tile is one k-tile of the plane layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BT;
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / WN, wn = wid % WN;

  // ---- LDS-DMA sources: thread t fills chunk (t % SLOTS) of rows (t / SLOTS) + RPI * j of every image.  An address is
  // a UNIFORM base (plane, k-tile, tile origin: SGPRs, advanced by SALU) plus a per-thread 32-bit byte offset that never
  // changes, and the LDS destination is uniform too, so issuing a tile costs a wave no VALU instruction: a wave whose
  // SIMD partner is issuing MFMAs gets its VALU slots late (measured: 2000 shader clocks for eight DMA instructions
  // with a v_readfirstlane + v_lshl_add_u64 each, against 590 when the partner is idle).
  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  unsigned off_a[IPI], off_w[IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    const int r = r0 + RPI * j;
    const int64_t ar = m0 + r < p.M ? r : p.M - 1 - m0;            // rows / columns past the edge: any valid row, the
    const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;          // epilogue never stores what they feed
    off_a[j] = (unsigned)(ar * BK + sp * 8) * 2u;
    off_w[j] = (unsigned)(br * BK + sp * 8) * 2u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  const char* a_tile = reinterpret_cast<const char*>(p.Ah + m0 * BK);
  const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * BK);
  const int64_t a_plane = p.M * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;      // bytes
  const int64_t a_kt = p.M * BK * 2, w_kt = (int64_t)p.Nout * BK * 2;                       // bytes per k-tile of rows
  auto issue = [&](int kt, int buf) {
    if ((DIAG == 3 || DIAG == 5) && kt > 1) return;
    const unsigned st = lds0 + buf * STAGE + wid * 1024;          // + lane * 16 by the hardware
#pragma unroll
    for (int im = 0; im < 4; ++im) {
      const char* sb = (im < 2 ? a_tile : w_tile) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) + kt * (im < 2 ? a_kt : w_kt);
#pragma unroll
      for (int j = 0; j < IPI; ++j) lds_dma16(sb, im < 2 ? off_a[j] : off_w[j], st + im * IMG + j * (NT * 16));
    }
  };

  // ---- fragment reads: lane l takes row (l & 31), 16-byte slot 2*ks + (l >> 5) of the wave's 32-row blocks
  const int fr = lane & 31, hi = lane >> 5;
  const int gl = BK == 64 ? ((fr >> 1) & 7) : (BK == 32 ? plane_swz(fr) : ((fr >> 3) & 1));
  const int a_row = (wm * TI * 32 + fr) * ROWB, b_row = (wn * 64 + fr) * ROWB;
  const int sx = (hi ^ gl) * 16;                                  // slot (2*ks + hi) ^ gl = (2*ks) ^ (hi ^ gl)

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  uint64_t diag_t[4] = {0, 0, 0, 0};
  auto multiply = [&](int buf) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int so = sx ^ (ks * 32);
      f16x8 a[TI][2], b[2][2];
      if (DIAG == 4 || DIAG == 5) {                 // no fragment reads: operands are whatever the registers hold
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) asm volatile("" : "=v"(a[i][q]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q) asm volatile("" : "=v"(b[j][q]));
      } else {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          a[i][q] = *reinterpret_cast<const f16x8*>(st + q * IMG + a_row + i * 32 * ROWB + so);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          b[j][q] = *reinterpret_cast<const f16x8*>(st + (2 + q) * IMG + b_row + j * 32 * ROWB + so);
      }
      if (DIAG == 2) {
#pragma unroll
        for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(a[i][0]), "v"(a[i][1]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[j][0]), "v"(b[j][1]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };

  if (NBUF >= 3) {
    // NBUF - 1 k-tiles of LDS-DMA in flight ACROSS the barriers: a counted vmcnt retires only the tile about to be
    // multiplied, and the barrier is the raw instruction (__syncthreads() would drain the queue: vmcnt(0))
    constexpr int GL = 4 * IPI;                       // LDS-DMA instructions per thread per k-tile
#pragma unroll
    for (int st = 0; st < NBUF - 1; ++st)
      if (st < nk) issue(st, st);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NBUF - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * GL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tiles: nothing younger is in flight
      __builtin_amdgcn_s_barrier();                   // tile kt is in LDS for every wave; tile kt-1's stage is free
      if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
      multiply(kt % NBUF);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's fragment reads are done before it
    }                                                 // arrives at the barrier that frees the stage
  } else if (NBUF == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();                 // every wave is done reading the stage
      issue(kt, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the tile has landed
      __syncthreads();
      multiply(0);
    }
  } else if (DIAG == 7) {              // where a wave's time goes: shader-clock sums per phase, written over Y[0][8*wid ..]
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0x0F70);
      const uint64_t ta = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      const uint64_t t2 = __builtin_amdgcn_s_memtime();
      multiply(kt & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint64_t t3 = __builtin_amdgcn_s_memtime();
      diag_t[0] += ta - t0;            // own LDS-DMA share not landed yet
      diag_t[1] += t1 - ta;            // waiting for the other waves at the barrier
      diag_t[2] += t2 - t1;            // issuing the next tile's LDS-DMA
      diag_t[3] += t3 - t2;            // fragment reads + MFMA issue
    }
  } else {
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt has landed
      __syncthreads();                 // tile kt has landed for every wave; stage (kt+1)&1 is free again
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      multiply(kt & 1);
    }
  }

  // ---- epilogue: undo the scales (row, then column: the product of the two could leave fp32's range), bias, raw Y,
  // fp64 column statistics, |Y|max for the next layer's operand scale
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);
  float ymax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cl = wn * 64 + j * 32 + fr;
    const int col = n0 + cl;
    const bool cok = col < p.Nout;
    const float bias = cok ? p.bias[col] : 0.f;
    const float iw = cok ? p.inv_w[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < p.M && cok) {
          const float y = fmaf(acc[i][j][r] * p.inv_a[row], iw, bias);
          p.Y[row * p.ldy + col] = y;
          ymax = fmaxf(ymax, fabsf(y));
          cs += y;
          cq += (double)y * y;
        }
      }
    }
    cs += __shfl_xor(cs, 32, 64);
    cq += __shfl_xor(cq, 32, 64);
    if (lane < 32) {
      colred[(wm * 2 + 0) * BT + cl] = cs;
      colred[(wm * 2 + 1) * BT + cl] = cq;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BT; i += NT) {
    const int which = i / BT, cl = i % BT, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BT + cl] + colred[(1 * 2 + which) * BT + cl]);
  }
  if (p.amax_y) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    __syncthreads();
    float* wmax = reinterpret_cast<float*>(smem);
    if (lane == 0) wmax[wid] = ymax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = wmax[0];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, wmax[w]);
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(m));
    }
  }
  if (DIAG == 7 && blockIdx.x == 0) {
    __syncthreads();
    if (lane == 0)
      for (int q = 0; q < 4; ++q) p.Y[wid * 4 + q] = (float)diag_t[q] / (float)nk;
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, BK = 32, two LDS stages, ONE barrier per k-tile placed in the MIDDLE of the tile's MFMAs, fragments
// double-buffered in registers.  In gemm_f16p_kernel every k-tile starts with all eight waves behind a barrier with
// empty fragment registers: the matrix pipes idle for a whole LDS round trip per tile.  Here a k-tile's two 16-deep
// steps alternate between two fragment sets:
//     issue reads  F1 <- (tile k, step 1)                      | the reads fly under the MFMAs on F0
//     24 MFMAs on F0 (tile k, step 0)
//     wait own reads + own LDS-DMA of tile k+1; BARRIER         | every wave is done reading tile k's stage
//     issue LDS-DMA tile k+2 -> the stage tile k leaves; issue reads F0 <- (tile k+1, step 0)
//     24 MFMAs on F1 (tile k, step 1)                           | the new reads and the DMA fly under these
// so after the barrier the waves have 24 MFMAs each queued with operands already in registers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm_f16p_mid_kernel(SplitGemmParams p, int tiles_m, int tiles_n) {
  constexpr int BT = 256, BK = 32, NT = 512, TI = 4, ROWB = BK * 2, IMG = BT * ROWB, STAGE = 4 * IMG;
  constexpr int SLOTS = BK / 8, RPI = NT / SLOTS, IPI = BT / RPI;       // 4 slots, 128 rows per instruction, 2 per image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BT;
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / 4, wn = wid % 4;

  // LDS-DMA sources as in gemm_f16p_kernel: uniform base + loop-invariant per-thread byte offset (lds_dma16)
  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  unsigned off_a[IPI], off_w[IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    const int r = r0 + RPI * j;
    const int64_t ar = m0 + r < p.M ? r : p.M - 1 - m0;
    const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;
    off_a[j] = (unsigned)(ar * BK + sp * 8) * 2u;
    off_w[j] = (unsigned)(br * BK + sp * 8) * 2u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  const char* a_tile = reinterpret_cast<const char*>(p.Ah + m0 * BK);
  const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * BK);
  const int64_t a_plane = p.M * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;      // bytes
  const int64_t a_kt = p.M * BK * 2, w_kt = (int64_t)p.Nout * BK * 2;
  auto issue_one = [&](int kt, int buf, int g) {                       // instruction g = 2 * image + half
    const int im = g >> 1, j = g & 1;
    const char* sb = (im < 2 ? a_tile : w_tile) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) + kt * (im < 2 ? a_kt : w_kt);
    lds_dma16(sb, im < 2 ? off_a[j] : off_w[j], lds0 + buf * STAGE + wid * 1024 + im * IMG + j * (NT * 16));
  };
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int g = 0; g < 8; ++g) issue_one(kt, buf, g);
  };
  const int fr = lane & 31, hi = lane >> 5;
  const int gl = plane_swz(fr);
  const int a_row = (wm * TI * 32 + fr) * ROWB, b_row = (wn * 64 + fr) * ROWB;
  const int sx = (hi ^ gl) * 16;

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f16x8 fa[2][TI][2], fb[2][2][2];                  // [fragment set][block][piece]
  auto read_frags = [&](int set, int buf, int ks) {
    const unsigned char* st = smem + buf * STAGE;
    const int so = sx ^ (ks * 32);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) fa[set][i][q] = *reinterpret_cast<const f16x8*>(st + q * IMG + a_row + i * 32 * ROWB + so);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) fb[set][j][q] = *reinterpret_cast<const f16x8*>(st + (2 + q) * IMG + b_row + j * 32 * ROWB + so);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][1], fb[set][j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][0], acc[i][j], 0, 0, 0);
      }
  };

  auto mfma3 = [&](int set, int i, int j) {
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][1], fb[set][j][0], acc[i][j], 0, 0, 0);
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][1], acc[i][j], 0, 0, 0);
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][0], acc[i][j], 0, 0, 0);
  };

  const int nk = p.K / BK;
  issue(0, 0);
  if (nk > 1) {
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * IPI) : "memory");    // tile 0 landed (tile 1 may still fly)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    read_frags(1, buf, 1);                                             // step 1 of this tile: flies under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    // this wave's reads of stage `buf` are done, and its share of tile kt+1 has landed in the other stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) read_frags(0, buf ^ 1, 0);                        // step 0 of the next tile
    __builtin_amdgcn_sched_barrier(0);
    // The eight LDS-DMA instructions of tile kt+2 (into the stage tile kt leaves) go out BETWEEN the MFMAs: a wave's
    // DMA issue takes 70-300 shader clocks per instruction (the CU's address unit takes the eight waves' instructions
    // at 64 B/clk), and a wave that issues them in one run keeps its matrix pipe idle for all of it.
    const bool more = kt + 2 < nk;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      mfma3(1, g >> 1, g & 1);
      if (more) issue_one(kt + 2, buf, g);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue (as gemm_f16p_kernel)
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);
  float ymax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cl = wn * 64 + j * 32 + fr;
    const int col = n0 + cl;
    const bool cok = col < p.Nout;
    const float bias = cok ? p.bias[col] : 0.f;
    const float iw = cok ? p.inv_w[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < p.M && cok) {
          const float y = fmaf(acc[i][j][r] * p.inv_a[row], iw, bias);
          p.Y[row * p.ldy + col] = y;
          ymax = fmaxf(ymax, fabsf(y));
          cs += y;
          cq += (double)y * y;
        }
      }
    }
    cs += __shfl_xor(cs, 32, 64);
    cq += __shfl_xor(cq, 32, 64);
    if (lane < 32) {
      colred[(wm * 2 + 0) * BT + cl] = cs;
      colred[(wm * 2 + 1) * BT + cl] = cq;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BT; i += NT) {
    const int which = i / BT, cl = i % BT, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BT + cl] + colred[(1 * 2 + which) * BT + cl]);
  }
  if (p.amax_y) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    __syncthreads();
    float* wmax = reinterpret_cast<float*>(smem);
    if (lane == 0) wmax[wid] = ymax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = wmax[0];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, wmax[w]);
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(m));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Ping-pong: the two waves of every SIMD alternate roles.  256 x 256 tile, BK = 32, two LDS stages, TWO barriers per
// k-tile.  Waves 0-3 (rows 0-127) and waves 4-7 (rows 128-255) share the SIMDs pairwise; in every phase one wave of a
// pair issues its whole k-tile of MFMAs (48, operands already in registers) while its partner reads ITS next k-tile of
// fragments out of LDS (24 ds_read_b128) -- the matrix pipe always has exactly one wave feeding it and never waits for
// an LDS round trip or a barrier release (MI355X_MICROARCH.md, 'Two waves per SIMD', item 9: lockstep partners).
//     phase A(t):  waves 0-3: MFMAs of tile t          waves 4-7: read tile t;       all: tile t+1 landed; barrier
//     phase B(t):  all: LDS-DMA tile t+2 -> stage of tile t (both halves have read it)
//                  waves 0-3: read tile t+1            waves 4-7: MFMAs of tile t;   all: barrier
// Fragment reads are inline asm (hipcc would put vmcnt(0) in front of every ds_read that follows an LDS-DMA), all waits
// are explicit.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm_f16p_pp_kernel(SplitGemmParams p, int tiles_m, int tiles_n) {
  constexpr int BT = 256, BK = 32, NT = 512, TI = 4, ROWB = BK * 2, IMG = BT * ROWB, STAGE = 4 * IMG;
  constexpr int SLOTS = BK / 8, RPI = NT / SLOTS, IPI = BT / RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BT;
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / 4, wn = wid % 4;
  const bool first_half = wm == 0;                                    // scalar: the two roles are scalar branches

  // LDS-DMA sources as in gemm_f16p_kernel: uniform base + loop-invariant per-thread byte offset (lds_dma16)
  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  unsigned off_a[IPI], off_w[IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    const int r = r0 + RPI * j;
    const int64_t ar = m0 + r < p.M ? r : p.M - 1 - m0;
    const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;
    off_a[j] = (unsigned)(ar * BK + sp * 8) * 2u;
    off_w[j] = (unsigned)(br * BK + sp * 8) * 2u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  const char* a_tile = reinterpret_cast<const char*>(p.Ah + m0 * BK);
  const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * BK);
  const int64_t a_plane = p.M * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;      // bytes
  const int64_t a_kt = p.M * BK * 2, w_kt = (int64_t)p.Nout * BK * 2;
  auto issue_one = [&](int kt, int g) {                                // instruction g = 2 * image + half
    const int im = g >> 1, j = g & 1;
    const char* sb = (im < 2 ? a_tile : w_tile) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) + kt * (im < 2 ? a_kt : w_kt);
    lds_dma16(sb, im < 2 ? off_a[j] : off_w[j], lds0 + (kt & 1) * STAGE + wid * 1024 + im * IMG + j * (NT * 16));
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int g = 0; g < 8; ++g) issue_one(kt, g);
  };
  const int fr = lane & 31, hi = lane >> 5;
  const int gl = plane_swz(fr);
  const int a_row = (wm * TI * 32 + fr) * ROWB, b_row = (wn * 64 + fr) * ROWB;
  const int sx = (hi ^ gl) * 16;

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f16x8 fa[2][TI][2], fb[2][2][2];                  // [16-deep step][block][piece]: a whole k-tile of fragments
  auto read_tile = [&](int kt, bool dma) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned so = (unsigned)(sx ^ (ks * 32));
      const unsigned sa = lds0 + (kt & 1) * STAGE + a_row + so;
      const unsigned sb = lds0 + (kt & 1) * STAGE + 2 * IMG + b_row + so;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[ks][i][q]) : "v"(sa), "n"(q * IMG + i * 32 * ROWB));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[ks][j][q]) : "v"(sb), "n"(q * IMG + j * 32 * ROWB));
    }
    if (dma) issue(kt + 1);                                            // behind the reads: they are what the wave waits for
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
  };
  auto mfma_tile = [&](int kt, bool dma) {                             // dma: one LDS-DMA instruction of tile kt+2 per six MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][1], fb[ks][j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][0], fb[ks][j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][0], fb[ks][j][0], acc[i][j], 0, 0, 0);
        }
        if (dma) issue_one(kt + 2, ks * 4 + i);
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // Every wave runs the SAME stream  { read tile t ; barrier ; MFMAs of tile t ; barrier }, waves 4-7 one phase behind
  // waves 0-3 (one extra barrier at their start, one at the others' end): while one partner of a SIMD pair computes, the
  // other reads.  Only the LDS-DMA issue / wait points differ between the halves (scalar branches around a few
  // instructions): a tile's stage is free once the LATER half has read it, and must have landed before the EARLIER half
  // reads it.  Global phase g: first half reads tile t at g = 2t and computes at 2t+1; second half at 2t+1 and 2t+2.
  //   first half : issues its share of tile t+1 at the start of reading tile t   (g = 2t),  waits for it behind the MFMAs of t
  //   second half: issues its share of tile t+2 at the start of the MFMAs of t   (g = 2t+2), waits for t+1 behind reading t
  const int nk = p.K / BK;
  issue(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): this wave's share of tile 0
  __builtin_amdgcn_s_barrier();
  if (!first_half) {
    if (nk > 1) issue(1);                                              // its share of tile 1 (g = 0)
    __builtin_amdgcn_s_barrier();                                      // the stagger
  }
  for (int kt = 0; kt < nk; ++kt) {
    // ---- read phase
    read_tile(kt, first_half && kt + 1 < nk);                          // ends with lgkmcnt(0)
    if (!first_half) __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): share of tile kt+1
    __builtin_amdgcn_s_barrier();
    // ---- compute phase
    mfma_tile(kt, !first_half && kt + 2 < nk);
    if (first_half) __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): share of tile kt+1
    __builtin_amdgcn_s_barrier();
  }
  if (first_half) __builtin_amdgcn_s_barrier();                        // pairs with the other half's last barrier

  // ---- epilogue (as gemm_f16p_kernel)
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);
  float ymax = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cl = wn * 64 + j * 32 + fr;
    const int col = n0 + cl;
    const bool cok = col < p.Nout;
    const float bias = cok ? p.bias[col] : 0.f;
    const float iw = cok ? p.inv_w[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < p.M && cok) {
          const float y = fmaf(acc[i][j][r] * p.inv_a[row], iw, bias);
          p.Y[row * p.ldy + col] = y;
          ymax = fmaxf(ymax, fabsf(y));
          cs += y;
          cq += (double)y * y;
        }
      }
    }
    cs += __shfl_xor(cs, 32, 64);
    cq += __shfl_xor(cq, 32, 64);
    if (lane < 32) {
      colred[(wm * 2 + 0) * BT + cl] = cs;
      colred[(wm * 2 + 1) * BT + cl] = cq;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BT; i += NT) {
    const int which = i / BT, cl = i % BT, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BT + cl] + colred[(1 * 2 + which) * BT + cl]);
  }
  if (p.amax_y) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    __syncthreads();
    float* wmax = reinterpret_cast<float*>(smem);
    if (lane == 0) wmax[wid] = ymax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = wmax[0];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, wmax[w]);
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(m));
    }
  }
}

static int launch_pp(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + 255) / 256), tiles_n = (p.Nout + 255) / 256;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)2 * 4 * 256 * 32 * 2;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_f16p_pp_kernel), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL(gemm_f16p_pp_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
  return MTMC_OK;
}

static int launch_mid(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + 255) / 256), tiles_n = (p.Nout + 255) / 256;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)2 * 4 * 256 * 32 * 2;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_f16p_mid_kernel), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL(gemm_f16p_mid_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
  return MTMC_OK;
}

template <int BT, int BK, int NBUF, int MINB, int DIAG = 0>
static int launch_variant(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + BT - 1) / BT), tiles_n = (p.Nout + BT - 1) / BT;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  size_t lds = (size_t)NBUF * 4 * BT * BK * 2;
  if (lds < (size_t)4 * BT * sizeof(double)) lds = (size_t)4 * BT * sizeof(double);
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_f16p_kernel<BT, BK, NBUF, MINB, DIAG>), 128 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL((gemm_f16p_kernel<BT, BK, NBUF, MINB, DIAG>), dim3(grid), dim3(BT * 2), lds, s, p, tiles_m, tiles_n);
  return MTMC_OK;
}

static int launch_lab(const SplitGemmParams& p, hipStream_t s, int variant) {
  if (p.K % 64 || p.K > 2048 || p.M < 1 || p.Nout < 1) return 1;
  switch (variant) {
    case 2: return launch_variant<128, 32, 1, 3>(p, s);
    case 3: return launch_variant<128, 32, 2, 2>(p, s);
    case 4: return launch_variant<256, 32, 1, 1>(p, s);
    case 8: return launch_variant<128, 32, 3, 1>(p, s);       // counted-vmcnt pipeline, two k-tiles in flight
    case 9: return launch_mid(p, s);                          // mid-tile barrier, fragments double-buffered
    case 11: return launch_gemm_presplit(p, s);               // the product kernel: 16x16x32 MFMAs, plain two-stage loop
    case 10: return launch_pp(p, s);                          // ping-pong: SIMD partners alternate MFMA / read roles
    // timing experiments, results are WRONG:
    case 12: return launch_variant<256, 32, 2, 1, 2>(p, s);   // no MFMAs
    case 13: return launch_variant<256, 32, 2, 1, 3>(p, s);   // no LDS-DMA inside the loop
    case 14: return launch_variant<256, 32, 2, 1, 4>(p, s);   // no fragment reads
    case 15: return launch_variant<256, 32, 2, 1, 5>(p, s);   // MFMAs and barriers only
    case 17: return launch_variant<256, 32, 2, 1, 7>(p, s);   // right results + per-phase shader-clock averages over Y[0][0..31]
    default: return launch_variant<256, 32, 2, 1>(p, s);
  }
}

}  // namespace mtmc

extern "C" int32_t mtmc_lab_linear_presplit_raw(const float* A, int64_t lda, const float* W, const float* bias, float* Y, int64_t M,
                                                int32_t K, int32_t N, void* work, uint64_t work_bytes, uint32_t* scratch,
                                                double* stats, int32_t variant, void* stream) {
  if (!A || !W || !bias || !Y || !work || !scratch || M < 1 || K < 64 || K % 64 || K > 2048 || N < 1) return MTMC_E_ARG;
  const uint64_t a_bytes = (uint64_t)M * K * 4, w_bytes = (uint64_t)N * K * 4;
  const uint64_t ia_off = a_bytes, wh_off = (ia_off + (uint64_t)M * 4 + 255) / 256 * 256, iw_off = wh_off + w_bytes;
  if (work_bytes < iw_off + (uint64_t)N * 4) return MTMC_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, 3 * mtmc::kAmaxRep * sizeof(uint32_t), s) != hipSuccess) return MTMC_E_HIP;
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return MTMC_E_HIP;
  unsigned char* wk = static_cast<unsigned char*>(work);
  if (variant >= 0) {
    mtmc::launch_split_rows(A, lda, M, K, wk, reinterpret_cast<float*>(wk + ia_off), s);
    mtmc::launch_split_rows(W, K, N, K, wk + wh_off, reinterpret_cast<float*>(wk + iw_off), s);
  } else {
    variant = -variant - 1;          // negative: the planes in `work` are reused (times the GEMM alone)
  }
  mtmc::SplitGemmParams g;
  g.Ah = reinterpret_cast<const _Float16*>(wk); g.inv_a = reinterpret_cast<const float*>(wk + ia_off);
  g.Wh = reinterpret_cast<const _Float16*>(wk + wh_off); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
  g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
  g.M = M; g.K = K; g.Nout = N;
  const int rc = mtmc::launch_lab(g, s, variant);
  if (rc != 0) return rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG;
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}
