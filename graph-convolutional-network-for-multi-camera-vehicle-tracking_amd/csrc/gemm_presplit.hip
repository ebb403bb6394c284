// First node-encoder layer of MANY-ROW graphs on PRE-SPLIT operands (reference models/mlp.py:15-27 via models/mpn.py:168;
// dispatch: presplit_layer0 -- >= 4096 rows of a 128x128-tile plan, K <= 2048): 1.10-1.15 ms against 1.45 ms for the
// in-loop kernel at 100k rows, and the split pass (0.30 ms) replaces the |.|max pass over x (0.14 ms) that kernel
// needs; per-row power-of-two scales instead of one per tensor.  The forward runs gemm_f16p_m16_kernel (variant 11); the
// other variants (tile shapes, counted-vmcnt pipelines, the mid-barrier and ping-pong kernels, timing experiments) stay
// reachable through mtmc_linear_presplit_raw for A/B (tools/presplit_time.py).
//
// gemm_bn_f16x3_kernel splits every fp32 operand element into its two fp16 pieces inside the k-loop: an A element
// Nout/128 times, a W element M/128 times, through VGPRs and ds_write (≈ 80 B/clk/CU).  For the one layer that
// dominates a many-row forward (x[M][2048] -> 1024, reference models/mlp.py:15-27 via models/mpn.py:168) the split is
// taken out of the loop:
//   split_rows_kernel   x -> (h1, h2) fp16 planes + one power-of-two scale per ROW (one pass: the row is in registers)
//   gemm_f16p_*_kernel  plain fp16 MFMA GEMM on the planes, tiles brought in by LDS-DMA (global_load_lds_dwordx4:
//                       no VGPRs, no ds_write), three products a1w1 + a1w2 + a2w1, scales undone per row / column
// Same representation error bound as the in-loop split (22 mantissa bits per operand); per-row scales only tighten it.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace mtmc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 h2_t __attribute__((ext_vector_type(2)));

// The planes are stored K-TILE-MAJOR and PRE-SWIZZLED: the eight halves k .. k+7 (k % 8 == 0) of row `row` of a plane
// of `rows` rows live at
//     ((k / kPlaneKT) * rows + row) * kPlaneKT + 8 * (((k % kPlaneKT) / 8) ^ ((row >> 2) & 3)),
// i.e. memory holds the LDS image itself (bank swizzle included; tile origins are multiples of 16 rows), so the
// [256 rows][32 halves] image a GEMM block stages per k-tile is ONE contiguous 16 KB run that LDS-DMA copies in lane
// order: every global_load_lds_dwordx4 reads 1 KB of consecutive addresses, whole 128-byte lines.  Row-major planes made each such instruction touch
// sixteen 64-byte half lines 2*K bytes apart and the GEMM ran at the rate the CUs' L1s could be fed in half lines:
// 1.28 ms against 0.94 ms at 100000 x 2048 x 1024.
constexpr int kPlaneKT = 32;

// One LDS-DMA instruction (64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024) in lane order) in the form that costs the
// issuing wave NO VALU instruction: uniform 64-bit base in SGPRs + a per-lane 32-bit byte offset that is loop-invariant.
// hipcc selects the 64-bit-VGPR-address form for __builtin_amdgcn_global_load_lds here (one v_lshl_add_u64 per
// instruction); the compiler does not count this instruction in vmcnt, so every wait on it is written out.
__device__ __forceinline__ void lds_dma16(const void* base, unsigned lane_off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds) : "memory");   // m0 is reserved: hipcc sets it right before each use of its own
}

// ------------------------------------------------------------------------------------------------
// Half a wave per row (a wave takes rows 2w and 2w+1, so that its stores fill whole 128-byte lines of the k-tile-major
// planes: the two rows' 64-byte segments of a k-tile are adjacent): |row|max -> scale 2^(14-e) (exact),
// h1 = rtz(x*s), h2 = rtz(x*s - h1)  (v_cvt_pkrtz_f16_f32, see the codegen note in gemm_bn.hip).
// K <= 2048, K % 8 == 0: a lane holds 8 consecutive floats per 256-column chunk.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int K,
                                                         _Float16* __restrict__ H, int64_t plane, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63, l = lane & 31;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const bool live = row < rows;
  const float* src = X + (live ? row : rows - 1) * ld;
  float4 v[8][2];
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (j * 32 + l) * 8;
    if (k < K) {
      v[j][0] = *reinterpret_cast<const float4*>(src + k);
      v[j][1] = *reinterpret_cast<const float4*>(src + k + 4);
    } else {
      v[j][0] = v[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j][0].x), fabsf(v[j][0].y)), fmaxf(fabsf(v[j][0].z), fabsf(v[j][0].w))));
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j][1].x), fabsf(v[j][1].y)), fmaxf(fabsf(v[j][1].z), fabsf(v[j][1].w))));
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (!live) return;
  int e = 0;
  if (m > 0.f && m < 3e38f) (void)frexpf(m, &e);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  const float s = ldexpf(1.f, 14 - e);
  if (l == 0) inv[row] = ldexpf(1.f, e - 14);
  _Float16* d1 = H + row * kPlaneKT;                     // + (k / kPlaneKT) * rows * kPlaneKT + swizzled slot
  _Float16* d2 = d1 + plane;
  const int g = (int)((row >> 2) & 3);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (j * 32 + l) * 8;
    if (k < K) {
      uint4 q1, q2;
      auto two = [&](float a, float b, unsigned& o1, unsigned& o2) {
        const float x0 = a * s, x1 = b * s;
        const h2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        const h2_t lo = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
        o1 = __builtin_bit_cast(unsigned, h);
        o2 = __builtin_bit_cast(unsigned, lo);
      };
      two(v[j][0].x, v[j][0].y, q1.x, q2.x);
      two(v[j][0].z, v[j][0].w, q1.y, q2.y);
      two(v[j][1].x, v[j][1].y, q1.z, q2.z);
      two(v[j][1].z, v[j][1].w, q1.w, q2.w);
      const int64_t o = (int64_t)(k / kPlaneKT) * rows * kPlaneKT + ((((k % kPlaneKT) >> 3) ^ g) << 3);
      *reinterpret_cast<uint4*>(d1 + o) = q1;
      *reinterpret_cast<uint4*>(d2 + o) = q2;
    }
  }
}

void launch_split_rows(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, hipStream_t s) {
  const int64_t blocks = (rows + 7) / 8;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, ld, rows, K,
                     static_cast<_Float16*>(H), rows * (int64_t)K, inv);
}

// ------------------------------------------------------------------------------------------------
// BT x BT tile (128: 4 waves as 2x2, 64x64 each; 256: 8 waves as 2x4, 128x64 each), BK columns per k-tile, NBUF stages.
// A stage holds four images [BT rows][BK halves]: A piece 1, A piece 2, W piece 1, W piece 2.  An image is filled by
// LDS-DMA in lane order (16-byte chunk c of the image lands at byte 16*c), so the bank swizzle is applied to the
// SOURCE address: chunk `sp` of row r holds the row's 16-byte slot sp ^ g(r); the fragment reads apply the same XOR.
//   BK = 64 (128-byte rows, two rows per 256-byte bank row):  g(r) = (r >> 1) & 7
//   BK = 32 ( 64-byte rows, four rows per bank row):          g(r) = (r >> 2) & 3
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
  const int gl = BK == 64 ? ((fr >> 1) & 7) : (BK == 32 ? ((fr >> 2) & 3) : ((fr >> 3) & 1));
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
  const int gl = (fr >> 2) & 3;
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
// The same 256 x 256 x 32 tile on v_mfma_f32_16x16x32_f16 (variant 11).  One instruction takes the whole k-tile
// (K = 32) of a 16 x 16 output block: per flop it moves half the accumulator registers of the 32x32x16 form and twice the
// operand registers, and the bare-loop probe (tools/hazard/mfma_probe.hip, bit 128) sustains 12 % more flops per second
// with it at the socket's power cap, where this GEMM runs.  A wave's 128 x 64 share is 8 x 4 blocks of
// 16 x 16 (32 accumulator quads = the same 128 registers); fragment reads: lane l takes row l % 16 of a 16-row block and
// the 16-byte slot l / 16 of its 64-byte image row (the stored swizzle makes every 16-lane group conflict-free here
// too); plain two-stage loop, the compiler places the reads.
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void gemm_f16p_m16_kernel(SplitGemmParams p, int tiles_m, int tiles_n) {
  constexpr int BT = 256, BK = 32, NT = 512, ROWB = BK * 2, IMG = BT * ROWB, STAGE = 4 * IMG;
  constexpr int SLOTS = BK / 8, RPI = NT / SLOTS, IPI = BT / RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BT;
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / 4, wn = wid % 4;

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
  const int64_t a_plane = p.M * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;
  const int64_t a_kt = p.M * BK * 2, w_kt = (int64_t)p.Nout * BK * 2;
  auto issue = [&](int kt, int buf) {
    const unsigned st = lds0 + buf * STAGE + wid * 1024;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
      const char* sb = (im < 2 ? a_tile : w_tile) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) + kt * (im < 2 ? a_kt : w_kt);
#pragma unroll
      for (int j = 0; j < IPI; ++j) lds_dma16(sb, im < 2 ? off_a[j] : off_w[j], st + im * IMG + j * (NT * 16));
    }
  };

  const int r16 = lane & 15, ks = lane >> 4;                       // fragment row inside a 16-row block, 8-half k group
  const int so = (ks ^ ((r16 >> 2) & 3)) * 16;                     // the stored swizzle: slot ^ ((row >> 2) & 3)
  const int a_row = (wm * 128 + r16) * ROWB + so, b_row = (wn * 64 + r16) * ROWB + so;

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](int buf) {
    const unsigned char* st = smem + buf * STAGE;
    f16x8 b[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(st + (2 + q) * IMG + b_row + j * 16 * ROWB);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f16x8 a[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) a[q] = *reinterpret_cast<const f16x8*>(st + q * IMG + a_row + i * 16 * ROWB);
      // the three products of a block go out back to back: the second and third take the accumulator the first just
      // produced (1.072 ms against 1.101 ms for runs of four MFMAs on four different accumulators, same box: the chained
      // form draws less power, and at the power cap that is time)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][1], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
    }
  };

  const int nk = p.K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of tile kt has landed
    __syncthreads();                      // ... for every wave; stage (kt+1)&1 is free again
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    multiply(kt & 1);
  }

  // ---- epilogue: a lane holds, per 16 x 16 block, rows 4*(lane/16) .. +3 of column lane%16 -- stored as they stand, a
  // wave-instruction would write four 64-byte row segments (half lines: WRITE_SIZE 1.40 x the bytes of Y at config 4).
  // The stages are free now: every wave transposes its 16 x 64 strips through a private 4 KB of LDS and writes whole
  // 256-byte row segments, 16 bytes per lane.  The column statistics are taken from the registers on the way.
  __syncthreads();
  constexpr int kStgLd = 64;                            // floats per staged row: conflict-free b128 reads (2-way b32 writes: free)
  float* stg = reinterpret_cast<float*>(smem) + wid * (16 * kStgLd);
  double* colred = reinterpret_cast<double*>(smem + 8 * 16 * kStgLd * sizeof(float));   // behind the eight staging strips
  const bool vec_ok = (p.ldy & 3) == 0 && ((uintptr_t)p.Y & 15) == 0;
  float ymax = 0.f;
  float bias[4], iw[4];
  double cs[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 64 + j * 16 + r16;
    const bool cok = col < p.Nout;
    bias[j] = cok ? p.bias[col] : 0.f;
    iw[j] = cok ? p.inv_w[col] : 0.f;
  }
  const int rrow = lane >> 4, rcol = (lane & 15) * 4;   // read-back: row rrow + 4q of the strip, columns rcol .. rcol + 3
  const int gcol = n0 + wn * 64 + rcol;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row0 = m0 + wm * 128 + i * 16;
    float ia[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ia[r] = p.inv_a[row0 + 4 * ks + r < p.M ? row0 + 4 * ks + r : p.M - 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool cok = n0 + wn * 64 + j * 16 + r16 < p.Nout;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = fmaf(acc[i][j][r] * ia[r], iw[j], bias[j]);
        stg[(4 * ks + r) * kStgLd + j * 16 + r16] = y;
        if (row0 + 4 * ks + r < p.M && cok) {
          ymax = fmaxf(ymax, fabsf(y));
          cs[j] += y;
          cq[j] += (double)y * y;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // (LDS executes a wave's accesses in order: this only
    __builtin_amdgcn_wave_barrier();                              //  keeps the compiler from reordering across lanes' data)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = rrow + 4 * q;
      const int64_t row = row0 + rr;
      const float4 v = *reinterpret_cast<const float4*>(stg + rr * kStgLd + rcol);
      if (row < p.M) {
        float* dst = p.Y + row * p.ldy + gcol;
        if (vec_ok && gcol + 3 < p.Nout) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (gcol + t < p.Nout) dst[t] = vv[t];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cl = wn * 64 + j * 16 + r16;
    double a = cs[j], b = cq[j];
    a += __shfl_xor(a, 16, 64);
    b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (lane < 16) {
      colred[(wm * 2 + 0) * BT + cl] = a;
      colred[(wm * 2 + 1) * BT + cl] = b;
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
  const int gl = (fr >> 2) & 3;
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

static int launch_m16(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + 255) / 256), tiles_n = (p.Nout + 255) / 256;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)2 * 4 * 256 * 32 * 2;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_f16p_m16_kernel), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL(gemm_f16p_m16_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
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

bool presplit_layer0(int64_t rows, int K, int Nout) {
  const bool off = knobs().gemm_no_presplit || knobs().gemm_fp32 || knobs().gemm_no_f16;
  int sk;
  return !off && K % 64 == 0 && K <= 2048 && rows >= 4096 && gemm_plan(rows, K, Nout, &sk) == 2;
}

int launch_gemm_presplit(const SplitGemmParams& p, hipStream_t s, int variant) {
  if (p.K % 64 || p.K > 2048 || p.M < 1 || p.Nout < 1) return 1;
  switch (variant) {
    case 2: return launch_variant<128, 32, 1, 3>(p, s);
    case 3: return launch_variant<128, 32, 2, 2>(p, s);
    case 4: return launch_variant<256, 32, 1, 1>(p, s);
    case 8: return launch_variant<128, 32, 3, 1>(p, s);       // counted-vmcnt pipeline, two k-tiles in flight
    case 9: return launch_mid(p, s);                          // mid-tile barrier, fragments double-buffered
    case 11: return launch_m16(p, s);                         // 16x16x32 MFMAs, plain two-stage loop
    case 10: return launch_pp(p, s);                          // ping-pong: SIMD partners alternate MFMA / read roles
    // timing experiments, results are WRONG (DESIGN.md 3.1 quotes them; tools/presplit_time.py runs them):
    case 12: return launch_variant<256, 32, 2, 1, 2>(p, s);   // no MFMAs
    case 13: return launch_variant<256, 32, 2, 1, 3>(p, s);   // no LDS-DMA inside the loop
    case 14: return launch_variant<256, 32, 2, 1, 4>(p, s);   // no fragment reads
    case 15: return launch_variant<256, 32, 2, 1, 5>(p, s);   // MFMAs and barriers only
    case 17: return launch_variant<256, 32, 2, 1, 7>(p, s);   // right results + per-phase shader-clock averages over Y[0][0..31]
    default: return launch_variant<256, 32, 2, 1>(p, s);
  }
}

}  // namespace mtmc
