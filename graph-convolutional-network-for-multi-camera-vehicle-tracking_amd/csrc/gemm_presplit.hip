// First node-encoder layer of MANY-ROW graphs on PRE-SPLIT operands (reference models/mlp.py:15-27 via models/mpn.py:168;
// dispatch: presplit_layer0 -- >= 4096 rows of a 128x128-tile plan, K <= 2048): 1.28 ms against 1.45 ms for the in-loop
// kernel at 100k rows, and the split pass (0.28 ms) replaces the |.|max pass over x (0.14 ms) that kernel needs; per-row
// power-of-two scales instead of one per tensor.  Variants other than the default (tile shapes, counted-vmcnt pipelines,
// the mid-barrier kernel) stay reachable through mtmc_linear_presplit_raw for A/B (tools/presplit_time.py).
//
// gemm_bn_f16x3_kernel splits every fp32 operand element into its two fp16 pieces inside the k-loop: an A element
// Nout/128 times, a W element M/128 times, through VGPRs and ds_write (≈ 80 B/clk/CU).  For the one layer that
// dominates a many-row forward (x[M][2048] -> 1024, reference models/mlp.py:15-27 via models/mpn.py:168) the split is
// taken out of the loop:
//   split_rows_kernel   x -> (h1, h2) fp16 planes + one power-of-two scale per ROW (one pass: the row is in registers)
//   gemm_f16p_kernel    plain fp16 MFMA GEMM on the planes, tiles brought in by LDS-DMA (global_load_lds_dwordx4:
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

// ------------------------------------------------------------------------------------------------
// One wave per row: |row|max -> scale 2^(14-e) (exact), h1 = rtz(x*s), h2 = rtz(x*s - h1)  (v_cvt_pkrtz_f16_f32, see
// the codegen note in gemm_bn.hip).  K <= 2048, K % 8 == 0: a lane holds 8 consecutive floats per 512-column chunk.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int K,
                                                         _Float16* __restrict__ H, int64_t plane, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = X + row * ld;
  float4 v[4][2];
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = (j * 64 + lane) * 8;
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
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  int e = 0;
  if (m > 0.f && m < 3e38f) (void)frexpf(m, &e);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  const float s = ldexpf(1.f, 14 - e);
  if (lane == 0) inv[row] = ldexpf(1.f, e - 14);
  _Float16* d1 = H + row * (int64_t)K;
  _Float16* d2 = d1 + plane;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = (j * 64 + lane) * 8;
    if (k < K) {
      uint4 q1, q2;
      auto two = [&](float a, float b, unsigned& o1, unsigned& o2) {
        const float x0 = a * s, x1 = b * s;
        const h2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        const h2_t l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
        o1 = __builtin_bit_cast(unsigned, h);
        o2 = __builtin_bit_cast(unsigned, l);
      };
      two(v[j][0].x, v[j][0].y, q1.x, q2.x);
      two(v[j][0].z, v[j][0].w, q1.y, q2.y);
      two(v[j][1].x, v[j][1].y, q1.z, q2.z);
      two(v[j][1].z, v[j][1].w, q1.w, q2.w);
      *reinterpret_cast<uint4*>(d1 + k) = q1;
      *reinterpret_cast<uint4*>(d2 + k) = q2;
    }
  }
}

void launch_split_rows(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, hipStream_t s) {
  const int64_t blocks = (rows + 3) / 4;
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
template <int BT, int BK, int NBUF, int MINB>
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
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BT;
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;

  // ---- LDS-DMA sources: thread t fills chunk (t % SLOTS) of rows (t / SLOTS) + RPI * j of every image
  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  const int gsrc = BK == 64 ? ((r0 >> 1) & 7) : (BK == 32 ? ((r0 >> 2) & 3) : ((r0 >> 3) & 1));   // RPI % 16 == 0: g(r0 + RPI*j) = g(r0)
  const int scol = (sp ^ gsrc) * 8;
  const _Float16* src[4][IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    const int r = r0 + RPI * j;
    const int64_t ar = m0 + r < p.M ? m0 + r : p.M - 1;           // rows / columns past the edge: any valid row, the
    const int64_t br = n0 + r < p.Nout ? n0 + r : p.Nout - 1;     // epilogue never stores what they feed
    src[0][j] = p.Ah + ar * p.K + scol;
    src[1][j] = src[0][j] + p.M * (int64_t)p.K;
    src[2][j] = p.Wh + br * p.K + scol;
    src[3][j] = src[2][j] + (int64_t)p.Nout * p.K;
  }
  auto issue = [&](int kt, int buf) {
    unsigned char* st = smem + buf * STAGE + wid * 1024;          // + lane * 16 by the hardware
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int j = 0; j < IPI; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[im][j] + kt * BK),
                                         (__attribute__((address_space(3))) void*)(st + im * IMG + j * (NT * 16)), 16, 0, 0);
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
  auto multiply = [&](int buf) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int so = sx ^ (ks * 32);
      f16x8 a[TI][2], b[2][2];
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
      __syncthreads();                 // waits vmcnt(0): the tile has landed, for every wave
      multiply(0);
    }
  } else {
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();                 // tile kt has landed (vmcnt(0) of every wave); stage (kt+1)&1 is free again
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
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / 4, wn = wid % 4;

  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  const int gsrc = (r0 >> 2) & 3;
  const int scol = (sp ^ gsrc) * 8;
  // two base pointers per thread (its row of the A / W tile); the second piece, the second 128-row half and the k-tile
  // are uniform offsets added at issue time -- eight resident 64-bit pointers would not fit beside the fragments
  const int64_t ar0 = m0 + r0 < p.M ? m0 + r0 : p.M - 1, ar1 = m0 + r0 + RPI < p.M ? m0 + r0 + RPI : p.M - 1;
  const int64_t br0 = n0 + r0 < p.Nout ? n0 + r0 : p.Nout - 1, br1 = n0 + r0 + RPI < p.Nout ? n0 + r0 + RPI : p.Nout - 1;
  const _Float16* a_base = p.Ah + ar0 * p.K + scol;
  const _Float16* w_base = p.Wh + br0 * p.K + scol;
  const int a_step = (int)(ar1 - ar0) * p.K, w_step = (int)(br1 - br0) * p.K;   // elements to the thread's second row
  const int64_t a_plane = p.M * (int64_t)p.K, w_plane = (int64_t)p.Nout * p.K;
  auto issue = [&](int kt, int buf) {
    unsigned char* st = smem + buf * STAGE + wid * 1024;
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int j = 0; j < IPI; ++j) {
        const _Float16* g = (im < 2 ? a_base : w_base) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) +
                            (j ? (im < 2 ? a_step : w_step) : 0) + kt * BK;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(st + im * IMG + j * (NT * 16)), 16, 0, 0);
      }
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
    if (kt + 2 < nk) issue(kt + 2, buf);                               // the stage tile kt leaves
    if (kt + 1 < nk) read_frags(0, buf ^ 1, 0);                        // step 0 of the next tile
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
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

static void launch_mid(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + 255) / 256), tiles_n = (p.Nout + 255) / 256;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)2 * 4 * 256 * 32 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16p_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_f16p_mid_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
}

template <int BT, int BK, int NBUF, int MINB>
static void launch_variant(const SplitGemmParams& p, hipStream_t s) {
  const int tiles_m = (int)((p.M + BT - 1) / BT), tiles_n = (p.Nout + BT - 1) / BT;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  size_t lds = (size_t)NBUF * 4 * BT * BK * 2;
  if (lds < (size_t)4 * BT * sizeof(double)) lds = (size_t)4 * BT * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16p_kernel<BT, BK, NBUF, MINB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_f16p_kernel<BT, BK, NBUF, MINB>), dim3(grid), dim3(BT * 2), lds, s, p, tiles_m, tiles_n);
}

bool presplit_layer0(int64_t rows, int K, int Nout) {
  static const bool off = getenv("MTMC_GEMM_NO_PRESPLIT") != nullptr || getenv("MTMC_GEMM_FP32") != nullptr || getenv("MTMC_GEMM_NO_F16") != nullptr;
  int sk;
  return !off && K % 64 == 0 && K <= 2048 && rows >= 4096 && gemm_plan(rows, K, Nout, &sk) == 2;
}

int launch_gemm_presplit(const SplitGemmParams& p, hipStream_t s, int variant) {
  if (p.K % 64 || p.K > 2048 || p.M < 1 || p.Nout < 1) return 1;
  switch (variant) {
    case 1: launch_variant<128, 64, 1, 2>(p, s); break;
    case 2: launch_variant<128, 32, 1, 3>(p, s); break;
    case 3: launch_variant<128, 32, 2, 2>(p, s); break;
    case 4: launch_variant<256, 32, 1, 1>(p, s); break;
    case 5: launch_variant<256, 64, 1, 1>(p, s); break;
    case 6: launch_variant<256, 16, 4, 1>(p, s); break;       // counted-vmcnt pipeline, three k-tiles in flight
    case 7: launch_variant<256, 16, 3, 1>(p, s); break;
    case 8: launch_variant<128, 32, 3, 1>(p, s); break;
    case 9: launch_mid(p, s); break;                          // mid-tile barrier, fragments double-buffered
    default: launch_variant<256, 32, 2, 1>(p, s); break;
  }
  return 0;
}

}  // namespace mtmc
