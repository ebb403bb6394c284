// First node-encoder layer of MANY-ROW graphs on PRE-SPLIT operands (reference models/mlp.py:15-27 via models/mpn.py:168;
// dispatch: presplit_layer0 -- >= 4096 rows of a 128x128-tile plan, K <= 2048): 1.10 ms against 1.45 ms for the
// in-loop kernel at 100k rows, and the split pass (0.30 ms) replaces the |.|max pass over x (0.14 ms) that kernel
// needs; per-row power-of-two scales instead of one per tensor.  This file holds what the forward runs: split_rows_kernel
// and gemm_f16p_m16_kernel.  The kernels it was chosen against (other tile shapes, counted-vmcnt pipelines, the mid-barrier
// and ping-pong schedules, timing experiments with deliberately wrong results) live in lab/presplit_lab.hip and are built
// into libmtmc_lab.so for tools/presplit_time.py and tests/test_gpu_gemm_presplit.py -- not into the product library.
//
// gemm_bn_f16x3_kernel splits every fp32 operand element into its two fp16 pieces inside the k-loop: an A element
// Nout/128 times, a W element M/128 times, through VGPRs and ds_write (~ 80 B/clk/CU).  For the one layer that
// dominates a many-row forward (x[M][2048] -> 1024) the split is taken out of the loop:
//   split_rows_kernel     x -> (h1, h2) fp16 planes + one power-of-two scale per ROW (one pass: the row is in registers)
//   gemm_f16p_m16_kernel  plain fp16 MFMA GEMM on the planes, tiles brought in by LDS-DMA (global_load_lds_dwordx4:
//                         no VGPRs, no ds_write), three products a1w1 + a1w2 + a2w1, scales undone per row / column
// Same representation error bound as the in-loop split (22 mantissa bits per operand); per-row scales only tighten it.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "lds_dma.h"
#include "split_body.h"

namespace mtmc {

// ------------------------------------------------------------------------------------------------
// Half a wave per row (a wave takes rows 2w and 2w+1, so that its stores fill whole 128-byte lines of the k-tile-major
// planes: the two rows' 64-byte segments of a k-tile are adjacent): |row|max -> scale 2^(14-e) (exact),
// h1 = rtz(x*s), h2 = rtz(x*s - h1)  (v_cvt_pkrtz_f16_f32, see the codegen note in gemm_bn.hip).
// K <= 2048, K % 8 == 0: a lane holds 8 consecutive floats per 256-column chunk.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int K,
                                                         _Float16* __restrict__ H, int64_t plane, float* __restrict__ inv,
                                                         int64_t r_lo, int64_t r_hi) {
  split_rows_body(X, ld, rows, K, H, plane, inv, r_lo, r_hi, (int)blockIdx.x, nullptr, nullptr);   // (split_body.h)
}

void launch_split_rows_range(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, int64_t r_lo, int64_t r_hi,
                             hipStream_t s) {
  if (r_hi <= r_lo) return;
  const int64_t blocks = (r_hi - r_lo + 7) / 8;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, ld, rows, K,
                     static_cast<_Float16*>(H), rows * (int64_t)K, inv, r_lo, r_hi);
}
void launch_split_rows(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, hipStream_t s) {
  launch_split_rows_range(X, ld, rows, K, H, inv, 0, rows, s);
}

// ------------------------------------------------------------------------------------------------
// The same 256 x 256 x 32 tile on v_mfma_f32_16x16x32_f16.  One instruction takes the whole k-tile
// (K = 32) of a 16 x 16 output block: per flop it moves half the accumulator registers of the 32x32x16 form and twice the
// operand registers, and the bare-loop probe (tools/hazard/mfma_probe.hip, bit 128) sustains 12 % more flops per second
// with it at the socket's power cap, where this GEMM runs.  A wave's 128 x 64 share is 8 x 4 blocks of
// 16 x 16 (32 accumulator quads = the same 128 registers); fragment reads: lane l takes row l % 16 of a 16-row block and
// the 16-byte slot l / 16 of its 64-byte image row (the stored swizzle makes every 16-lane group conflict-free here
// too); plain two-stage loop, the compiler places the reads.
// ------------------------------------------------------------------------------------------------
// The tile HEIGHT is a launch parameter (presplit_tile_rows): the first row of waves always takes 128 rows, the second
// bm - 128 (a multiple of 16), and the two waves that share a SIMD are one of each -- with 256-row tiles the last round of
// workgroups of config 4 (1564 tiles = 6.11 rounds on 256 CUs) ran on a tenth of the chip.
__global__ __launch_bounds__(512, 1) void gemm_f16p_m16_kernel(SplitGemmParams p, int tiles_m, int tiles_n, int bm) {
  constexpr int BT = 256, BK = 32, NT = 512, ROWB = BK * 2, IMG = BT * ROWB, STAGE = 4 * IMG;
  constexpr int SLOTS = BK / 8, RPI = NT / SLOTS, IPI = BT / RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = p.m_lo + (int64_t)tm_idx * bm;                 // (m_lo: first row of this launch's panel)
  const int64_t Mt = p.M_rows > 0 ? p.M_rows : p.M;                 // rows the planes are laid out for
  const int n0 = tn_idx * BT;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / 4, wn = wid % 4;
  const int nblk = wm == 0 ? 8 : (bm - 128) / 16;                  // 16-row blocks of this wave
  const int64_t m_end = m0 + bm < p.M ? m0 + bm : p.M;             // rows of this tile: [m0, m_end)

  const int r0 = threadIdx.x / SLOTS, sp = threadIdx.x % SLOTS;
  unsigned off_a[IPI], off_w[IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    const int r = r0 + RPI * j;
    const int64_t ar = m0 + r < m_end ? r : m_end - 1 - m0;       // (rows past the tile: a valid row again, never used)
    const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;
    off_a[j] = (unsigned)(ar * BK + sp * 8) * 2u;
    off_w[j] = (unsigned)(br * BK + sp * 8) * 2u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  const char* a_tile = reinterpret_cast<const char*>(p.Ah + m0 * BK);
  const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * BK);
  const int64_t a_plane = Mt * (int64_t)p.K * 2, w_plane = (int64_t)p.Nout * p.K * 2;
  const int64_t a_kt = Mt * BK * 2, w_kt = (int64_t)p.Nout * BK * 2;
  auto issue = [&](int kt, int buf) {
    const unsigned st = lds0 + buf * STAGE + wid * 1024;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
      const char* sb = (im < 2 ? a_tile : w_tile) + (im & 1 ? (im < 2 ? a_plane : w_plane) : 0) + kt * (im < 2 ? a_kt : w_kt);
#pragma unroll
      for (int j = 0; j < IPI; ++j) {
        // (A rows 128 + 16 wid .. + 15 of the second pass lie past a shorter tile's height: wave-uniform skip)
        if (im < 2 && j == 1 && 128 + wid * 16 >= bm) continue;
        lds_dma16(sb, im < 2 ? off_a[j] : off_w[j], st + im * IMG + j * (NT * 16));
      }
    }
  };

  const int r16 = lane & 15, ks = lane >> 4;                       // fragment row inside a 16-row block, 8-half k group
  const int so = (ks ^ plane_swz(r16)) * 16;                       // the stored swizzle (lds_dma.h)
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
      if (i < nblk) {                                      // (blocks past the tile's height: wave-uniform skip)
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
    if (i < nblk) {                                       // (no `break`: the loop must unroll, acc[] is registers)
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
        if (row0 + 4 * ks + r < m_end && cok) {
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
      if (row < m_end) {
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

bool presplit_layer0(int64_t rows, int K, int Nout) {
  const bool off = knobs().gemm_no_presplit || knobs().gemm_fp32 || knobs().gemm_no_f16;
  int sk;
  return !off && K % 64 == 0 && K <= 2048 && rows >= 4096 && gemm_plan(rows, K, Nout, &sk) == 2;
}

// Tile height: 144 .. 256 rows in steps of 16; the one that minimises rounds of workgroups x (height + a tile's fixed
// cost in row equivalents).  Config 4: 224 rows, 1788 tiles = 6.98 rounds of 7/8 of the work each, instead of 1564 tiles =
// 6.11 rounds paid as 7.
int presplit_tile_rows(int64_t M, int tiles_n) {
  const int forced = knobs().presplit_rows;
  if (forced >= 144 && forced <= 256 && forced % 16 == 0) return forced;
  int best = 256;
  int64_t best_cost = -1;
  for (int bm = 256; bm >= 144; bm -= 16) {
    const int64_t tiles = (M + bm - 1) / bm * tiles_n, rounds = (tiles + 255) / 256;
    // + 36: a tile's fixed cost in row equivalents (its W operand, prologue, epilogue), fitted to config 5 on one box:
    // 256 / 240 / 224 rows = 62 / 66 / 70 rounds took 10474 / 10452 / 10527 us
    const int64_t cost = rounds * (bm + 36);
    if (best_cost < 0 || cost < best_cost) { best = bm; best_cost = cost; }
  }
  return best;
}

int launch_gemm_presplit(const SplitGemmParams& p, hipStream_t s) {
  if (p.K % 64 || p.K > 2048 || p.M < 1 || p.Nout < 1 || p.m_lo < 0 || p.m_lo >= p.M || (p.m_lo & 15) ||
      (p.M_rows > 0 && p.M_rows < p.M))
    return 1;
  const int tiles_n = (p.Nout + 255) / 256;
  const int bm = (p.bm >= 144 && p.bm <= 256 && p.bm % 16 == 0) ? p.bm : presplit_tile_rows(p.M - p.m_lo, tiles_n);
  const int tiles_m = (int)((p.M - p.m_lo + bm - 1) / bm);
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)2 * 4 * 256 * 32 * 2;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_f16p_m16_kernel), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL(gemm_f16p_m16_kernel, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n, bm);
  return MTMC_OK;
}

}  // namespace mtmc
