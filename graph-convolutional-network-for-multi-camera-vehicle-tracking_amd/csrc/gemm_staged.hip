// Node-encoder layers 1.. of MANY-ROW graphs (reference models/mlp.py:14-27 via models/mpn.py:131): Y = relu(bn(Y_prev)) . W^T + b
// with the operand conversion taken OFF the matrix waves.
//
// gemm_bn_f16x3_kernel (gemm_bn.hip) does everything in every wave: global load -> BatchNorm affine + ReLU -> two-piece fp16
// split -> ds_write -> barrier -> ds_read -> MFMA.  Conversion and staging alternate with the matrix work instead of
// overlapping it: layer 1 of config 4 (100000 x 1024 -> 512) ran at 0.26 of its bound.  Here a 512-thread workgroup is
// split by ROLE (the SIMDs each host one wave of either kind: MI355X_MICROARCH.md, wave placement 0->2->1->3):
//   waves 0-3  PRODUCERS  A: global load (fp32 raw Y_prev, one k-tile ahead in registers) -> affine + ReLU -> scale ->
//                            fp16 pieces (v_cvt_pkrtz) -> ds_write_b128 into the NEXT stage, in the swizzled image the
//                            fragment reads want;
//                         W: LDS-DMA of the layer's PRE-SPLIT weight planes (split_rows_kernel, once per forward: the same
//                            [rows][32] k-tile-major swizzled image as layer 0's operands) two k-tiles ahead;
//   waves 4-7  CONSUMERS  ds_read_b128 fragments + v_mfma_f32_32x32x16_f16, three products per fp32 product (a1w0 + a0w1 + a0w0);
// one s_barrier per k-tile.  The VALU work of a producer runs in the issue slots the consumer's MFMAs leave free
// (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles), the weight bytes never touch a VGPR, and an A element is
// converted Nout / BN times instead of Nout / 128.
// Tile: 128 rows x BN columns (BN = 256: layer 1; 128: layer 2), BK = 32; consumers 2 x 2, 64 x BN/2 each.
// Accuracy: the same two-piece split as the other encoder kernels (22 mantissa bits per operand);
// A scale: one power of two per launch from |Y_prev|max and the BatchNorm affine (as gemm_bn_f16x3_kernel), W: one per row.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace mtmc {

typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef __fp16 h2s_t __attribute__((ext_vector_type(2)));

constexpr int kSgBM = 128, kSgBK = 32, kSgRowB = kSgBK * 2;      // bytes per image row
constexpr int kSgAImg = kSgBM * kSgRowB;                         // one A piece of one stage: 8 KB
constexpr int kSgNA = 2, kSgNW = 3;                              // stages: A double-buffered, W three deep

// (as gemm_presplit.hip: the LDS-DMA form that costs the issuing wave no VALU instruction; the compiler does not count
// it in vmcnt, every wait on it is written out)
__device__ __forceinline__ void sg_lds_dma16(const void* base, unsigned lane_off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds) : "memory");
}

template <int BN>
__global__ __launch_bounds__(512, 1) void gemm_staged_kernel(StagedGemmParams p, int tiles_m, int tiles_n) {
  constexpr int WIMG = BN * kSgRowB;                   // one W piece of one stage
  constexpr int TJ = BN / 64;                          // 32-column blocks per consumer wave (its share: 64 rows x BN/2 columns)
  constexpr int WJ = BN / 64;                          // DMA instructions per W piece per producer wave pass (64 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* w_st = smem;                                        // [kSgNW][2][BN][64 B]
  unsigned char* a_st = smem + kSgNW * 2 * WIMG;                     // [kSgNA][2][128][64 B]
  float* s_in = reinterpret_cast<float*>(a_st + kSgNA * 2 * kSgAImg);   // [K]
  float* t_in = s_in + p.K;                                          // [K]
  float* sc = t_in + p.K;                                            // [4]: scale of A, -, 1 / scale of A, -
  float* wred = sc + 4;                                              // [16]

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * kSgBM;
  const int n0 = tn_idx * BN;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wid < 4;

  // ---- prologue (everyone): BatchNorm affine of the K input columns, the bound on |relu(bn(.))| -> the A scale
  {
    float ms = 0.f, mt = 0.f;
    for (int kk = threadIdx.x; kk < p.K; kk += 512) {
      float sv, tv;
      bn_affine(p.stats_in[kk], p.stats_in[p.K + kk], p.count, p.gamma_in[kk], p.beta_in[kk], sv, tv);
      s_in[kk] = sv; t_in[kk] = tv;
      ms = fmaxf(ms, fabsf(sv));
      mt = fmaxf(mt, fabsf(tv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ms = fmaxf(ms, __shfl_xor(ms, off, 64));
      mt = fmaxf(mt, __shfl_xor(mt, off, 64));
    }
    if (lane == 0) { wred[wid] = ms; wred[8 + wid] = mt; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned ua = 0;
#pragma unroll
    for (int r = 0; r < kAmaxRep; ++r) ua = max(ua, p.amax_a[r]);
    float s8 = 0.f, t8 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { s8 = fmaxf(s8, wred[w]); t8 = fmaxf(t8, wred[8 + w]); }
    const float bound = fmaf(__uint_as_float(ua), s8, t8);
    int ea = 0;
    if (bound > 0.f && bound < 3e38f) (void)frexpf(bound, &ea);
    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
    sc[0] = ldexpf(1.f, 14 - ea);
    sc[2] = ldexpf(1.f, ea - 14);
  }
  __syncthreads();
  const float sa = sc[0];
  const int nk = p.K / kSgBK;
  const unsigned lds0 = (unsigned)(size_t)smem;

  if (producer) {
    // ---- A: lane t takes 16-byte slot (t & 3) (k = 8 * slot .. + 7 of the k-tile) of rows (t >> 2) and (t >> 2) + 64
    const int pt = threadIdx.x, sp = pt & 3, r0 = pt >> 2;
    const float* a_src[2];
    unsigned a_dst[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = r0 + 64 * h;
      const int64_t row = m0 + r < p.M ? m0 + r : p.M - 1;           // rows past M: any valid row (never stored)
      a_src[h] = p.A + row * p.lda + sp * 8;
      a_dst[h] = (unsigned)(r * kSgRowB + ((sp ^ ((r >> 2) & 3)) << 4));
    }
    // ---- W: thread t fills chunk (t & 3) of rows (t >> 2) + 64 j of both pieces: uniform base + loop-invariant lane offset
    unsigned off_w[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int r = r0 + 64 * j;
      const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;
      off_w[j] = (unsigned)(br * kSgBK + sp * 8) * 2u;
    }
    const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * kSgBK);
    const int64_t w_plane = (int64_t)p.Nout * p.K * 2, w_kt = (int64_t)p.Nout * kSgBK * 2;
    auto dma_w = [&](int kt) {
      const unsigned st = lds0 + (kt % kSgNW) * 2 * WIMG + wid * 1024;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const char* sb = w_tile + q * w_plane + kt * w_kt;
#pragma unroll
        for (int j = 0; j < WJ; ++j) sg_lds_dma16(sb, off_w[j], st + q * WIMG + j * 4096);
      }
    };
    float4 ra[2][2][2];                                  // [register set][row half][float4 of the 8]
    auto load_a = [&](int kt, int set) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[set][h][0] = *reinterpret_cast<const float4*>(a_src[h] + kt * kSgBK);
        ra[set][h][1] = *reinterpret_cast<const float4*>(a_src[h] + kt * kSgBK + 4);
      }
    };
    auto convert_a = [&](int kt, int set) {
      unsigned char* st = a_st + (kt & 1) * 2 * kSgAImg;
      const float4 s0 = *reinterpret_cast<const float4*>(s_in + kt * kSgBK + sp * 8);
      const float4 s1 = *reinterpret_cast<const float4*>(s_in + kt * kSgBK + sp * 8 + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(t_in + kt * kSgBK + sp * 8);
      const float4 t1 = *reinterpret_cast<const float4*>(t_in + kt * kSgBK + sp * 8 + 4);
      const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 v0 = ra[set][h][0], v1 = ra[set][h][1];
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint4 q1, q2;
        unsigned* o1 = reinterpret_cast<unsigned*>(&q1);
        unsigned* o2 = reinterpret_cast<unsigned*>(&q2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // relu(bn(y)) scaled into fp16's range, then h1 = rtz(x), h2 = rtz(x - h1) (exact residual; see gemm_bn.hip `put`)
          const float xa = fmaxf(fmaf(x[2 * e], sv[2 * e], tv[2 * e]), 0.f) * sa;
          const float xb = fmaxf(fmaf(x[2 * e + 1], sv[2 * e + 1], tv[2 * e + 1]), 0.f) * sa;
          const h2s_t hi = __builtin_amdgcn_cvt_pkrtz(xa, xb);
          const h2s_t lo = __builtin_amdgcn_cvt_pkrtz(xa - (float)hi[0], xb - (float)hi[1]);
          o1[e] = __builtin_bit_cast(unsigned, hi);
          o2[e] = __builtin_bit_cast(unsigned, lo);
        }
        *reinterpret_cast<uint4*>(st + a_dst[h]) = q1;
        *reinterpret_cast<uint4*>(st + kSgAImg + a_dst[h]) = q2;
      }
    };

    // One k-tile of producer work.  A(k) lives in register set k & 1: tile kt+2 is loaded into set `ld` = kt & 1 (free:
    // A(kt) went to LDS one k-tile ago) while tile kt+1, loaded a whole k-tile ago, is converted out of set `cv`.
    // ld / cv are literals at the call sites, so the register arrays are indexed statically after inlining.
    auto step = [&](int kt, int ld, int cv) {
      const bool more2 = kt + 2 < nk;
      if (more2) load_a(kt + 2, ld);
      if (kt + 1 < nk) convert_a(kt + 1, cv);            // -> A stage (kt+1)&1: the consumers left it at the last barrier
      if (more2) {
        dma_w(kt + 2);                                   // -> W stage (kt+2)%3 = (kt-1)%3: free since the last barrier
        // W(kt+1) must have landed before the barrier: younger than it are this k-tile's 4 loads and 2*WJ DMA instructions
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 + 2 * WJ) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    };
    // pipeline fill: W(0), W(1) by DMA; A(0) converted into stage 0; A(1) in registers
    dma_w(0);
    if (nk > 1) dma_w(1);
    load_a(0, 0);
    if (nk > 1) load_a(1, 1);
    convert_a(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt += 2) {
      step(kt, 0, 1);
      if (kt + 1 < nk) step(kt + 1, 1, 0);
    }
  }

  // ---- consumers: 2 x 2 waves, 64 rows x BN/2 columns each
  const int cw = wid & 3, wm = cw >> 1, wn = cw & 1;
  const int fr = lane & 31, hi = lane >> 5;
  f32x16s acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (!producer) {
    const int gl = (fr >> 2) & 3;
    const int a_row = (wm * 64 + fr) * kSgRowB, b_row = (wn * (BN / 2) + fr) * kSgRowB;
    const int sx = (hi ^ gl) * 16;
    __builtin_amdgcn_s_barrier();                        // pairs with the producers' pipeline-fill barrier
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* as = a_st + (kt & 1) * 2 * kSgAImg;
      const unsigned char* ws = w_st + (kt % kSgNW) * 2 * WIMG;
#pragma unroll
      for (int ks = 0; ks < kSgBK / 16; ++ks) {
        const int so = sx ^ (ks * 32);
        f16x8s a[2][2], b[TJ][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) a[i][q] = *reinterpret_cast<const f16x8s*>(as + q * kSgAImg + a_row + i * 32 * kSgRowB + so);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8s*>(ws + q * WIMG + b_row + j * 32 * kSgRowB + so);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads are done: the stage may be refilled
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue (consumers; the producers only join the barriers): undo the scales, bias, raw Y -- a 32 x 32 accumulator
  // register is two whole 128-byte row segments per wave-instruction --, fp64 column statistics, |Y|max
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);     // [2 (wm)][2 (sum, sq)][BN]
  float ymax = 0.f;
  if (!producer) {
    const float inv_a = sc[2];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int cl = wn * (BN / 2) + j * 32 + fr;
      const int col = n0 + cl;
      const bool cok = col < p.Nout;
      const float bias = cok ? p.bias[col] : 0.f;
      const float iw = cok ? p.inv_w[col] : 0.f;
      double cs = 0, cq = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < p.M && cok) {
            const float y = fmaf(acc[i][j][r] * inv_a, iw, bias);
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
        colred[(wm * 2 + 0) * BN + cl] = cs;
        colred[(wm * 2 + 1) * BN + cl] = cq;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += 512) {
    const int which = i / BN, cl = i % BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BN + cl] + colred[(1 * 2 + which) * BN + cl]);
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
      for (int w = 1; w < 8; ++w) m = fmaxf(m, wmax[w]);
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(m));
    }
  }
}

// which layers: eval mode (no Dropout here), an input BatchNorm (layers >= 1), many rows (the plan of the in-loop kernel
// would be 128 x 128 tiles), K a multiple of 32 whose affine fits beside the stages, Nout a multiple of 128
bool staged_layer(int64_t rows, int K, int Nout) {
  const Knobs& kn = knobs();
  if (kn.gemm_no_staged || kn.gemm_fp32 || kn.gemm_no_f16) return false;
  int sk;
  return rows >= 4096 && K % 32 == 0 && K >= 64 && K <= 2048 && Nout >= 128 && Nout % 128 == 0 && gemm_plan(rows, K, Nout, &sk) == 2;
}

int launch_gemm_staged(const StagedGemmParams& p, hipStream_t s) {
  if (p.M < 1 || p.K % 32 || p.K > 2048 || p.Nout % 128 || !p.stats_in || !p.amax_a) return 1;
  const int tiles_m = (int)((p.M + kSgBM - 1) / kSgBM);
  const bool wide = p.Nout % 256 == 0;
  const int bn = wide ? 256 : 128, tiles_n = p.Nout / bn;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)kSgNW * 2 * bn * kSgRowB + (size_t)kSgNA * 2 * kSgAImg + (size_t)(2 * p.K + 4 + 16) * sizeof(float);
  if (wide) {
    if (!allow_big_lds(reinterpret_cast<const void*>(gemm_staged_kernel<256>), 160 * 1024)) return MTMC_E_HIP;
    hipLaunchKernelGGL(gemm_staged_kernel<256>, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
  } else {
    if (!allow_big_lds(reinterpret_cast<const void*>(gemm_staged_kernel<128>), 160 * 1024)) return MTMC_E_HIP;
    hipLaunchKernelGGL(gemm_staged_kernel<128>, dim3(grid), dim3(512), lds, s, p, tiles_m, tiles_n);
  }
  return MTMC_OK;
}

}  // namespace mtmc
