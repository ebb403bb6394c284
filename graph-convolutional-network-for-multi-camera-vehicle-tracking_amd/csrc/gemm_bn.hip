// Node-encoder layer on the gfx950 matrix cores:   Y = act(A) . W^T + b,   column statistics of Y fused.
//
// One layer of reference MLP.forward (models/mlp.py:11-33) for the node encoder 2048->1024->512->128->32
// (models/mpn.py:131).  The reference runs Linear, BatchNorm1d (batch statistics), ReLU as three passes
// over [N, d]; here
//   * the previous layer's BatchNorm+ReLU is applied while its raw output is staged into LDS
//     (act(a) = relu(s_k a + t_k), s/t from that layer's fp64 column statistics), and
//   * this layer's column sum / sum of squares are reduced in fp64 in the epilogue,
// so every [N, d] activation is written once and read once.
//
// Arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fmaf chain; there is no
// reduced-precision f32 path on gfx950 and BatchNorm amplifies input error).  Large problems
// take gemm_bn_bf16x6_kernel below instead: the same layer as six bf16 products per fp32 product, at fp32 accuracy.
// Tiling: 256 threads = 2x2 waves, each wave TM x TN MFMA tiles of 32x32, BK = 32.  A and W tiles are both
// k-contiguous in HBM and in LDS (rows padded to 36 floats => conflict-free ds_read_b128); a lane's
// 16-byte read feeds four consecutive MFMAs (lane half h supplies k = 8*kk + 4*h + j in step j, for A and
// W alike, so the k-permutation cancels).  Register-staged double buffering: the next k-tile's global loads
// are in flight while the current one is multiplied.
#include "kernels.h"
#include <stdlib.h>
#include <mutex>
#include <set>
#include <utility>

namespace mtmc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// BK = 32 for the big-tile configuration (LDS budget), 64 for the few-row split-K configuration, where the
// k-loop is a chain of dependent HBM/L2 round trips and a deeper tile halves their number.
template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm_bn_kernel(GemmParams p, int tiles_m, int tiles_n) {
  drop_resolve(p.drop_in);
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDK = BK + 4;                        // row stride 144 B / 272 B: conflict-free ds_read_b128
  constexpr int C4 = BK / 4;                         // float4 per tile row
  constexpr int A4 = BM * C4 / 256, B4 = BN * C4 / 256;   // float4 loads per thread per k-tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [BM][LDK]
  float* Bs = smem + BM * LDK;                        // [BN][LDK]
  const int kc = p.K / p.split_k;                     // this block's share of K (split-K along blockIdx.y)
  const int k_base = blockIdx.y * kc;
  float* s_in = Bs + BN * LDK;                        // [kc]
  float* t_in = s_in + kc;                            // [kc]

  // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch); give each XCD
  // whole row panels so the A panel is re-read from its own L2 by the panel's column tiles.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BM;
  const int n0 = tn_idx * BN;

  const bool act = p.stats_in != nullptr;
  if (act) {
    for (int kk = threadIdx.x; kk < kc; kk += 256)
      bn_affine(p.stats_in[k_base + kk], p.stats_in[p.K + k_base + kk], p.count, p.gamma_in[k_base + kk],
                p.beta_in[k_base + kk], s_in[kk], t_in[kk]);
  }

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  float4 ra[A4], rb[B4];
  auto load_tiles = [&](int kt) {
    const int k0 = k_base + kt * BK;
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      const int64_t row = m0 + r;
      ra[i] = row < p.M ? *reinterpret_cast<const float4*>(p.A + row * p.lda + k0 + c4 * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      const int n = n0 + r;
      rb[i] = n < p.Nout ? *reinterpret_cast<const float4*>(p.W + (int64_t)n * p.K + k0 + c4 * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tiles = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      float4 v = ra[i];
      if (act) {
        const float4 s = *reinterpret_cast<const float4*>(s_in + k0 + c4 * 4);
        const float4 t = *reinterpret_cast<const float4*>(t_in + k0 + c4 * 4);
        v.x = fmaxf(fmaf(v.x, s.x, t.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s.y, t.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s.z, t.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s.w, t.w), 0.f);
        if (p.drop_in.on) {
          const unsigned long long idx = (unsigned long long)(m0 + r) * p.K + k_base + k0 + c4 * 4;
          float dv[4] = {v.x, v.y, v.z, v.w};
          drop_apply4(p.drop_in, p.drop_stream, idx, dv);        // (idx % 4 == 0: one hash for the four)
          v = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
      }
      *reinterpret_cast<float4*>(As + r * LDK + c4 * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      *reinterpret_cast<float4*>(Bs + r * LDK + c4 * 4) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = kc / BK;
  load_tiles(0);
  const int a_off = (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
  const int b_off = (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                 // s_in/t_in ready (kt == 0); previous tile fully consumed
    store_tiles(kt);
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(As + a_off + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + b_off + j * 32 * LDK + kk * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  if (p.split_k > 1) {
    // split-K: plain-store the partial tile into this slice's slab; combine_stats_kernel finishes the layer
    float* slab = p.slab + (size_t)blockIdx.y * p.M * p.Nout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.Nout) slab[row * p.Nout + col] = acc[i][j][r];
        }
    }
    return;
  }
  // epilogue: + bias, store raw Y, fp64 column statistics
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);   // [2 (wm)][2 (sum,sq)][BN]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = wn * TN * 32 + j * 32 + (lane & 31);
    const int col = n0 + cl;
    const float bias = col < p.Nout ? p.bias[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M && col < p.Nout) {
          const float y = acc[i][j][r] + bias;
          p.Y[row * p.ldy + col] = y;
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
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += 256) {
    const int which = i / BN, cl = i % BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BN + cl] + colred[(1 * 2 + which) * BN + cl]);
  }
}

// ------------------------------------------------------------------------------------------------
// The same layer on the bf16 matrix cores at fp32 accuracy: every fp32 operand is split into three bf16 pieces
// (x = x1 + x2 + x3, 24 mantissa bits in all) and the six products whose weight is >= 2^-16 are accumulated in fp32:
//     a.w ~= a1w1 + a1w2 + a2w1 + a2w2 + a1w3 + a3w1            (dropped terms <= 2^-24 |a||w|, i.e. fp32 rounding)
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K=16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 16-deep
// step: 2.7x fewer matrix cycles.  Measured end-to-end error at the logits: 1.4e-6 (fp32 MFMA: 3.7e-6; a 3-product
// split would be 5.5e-5), DESIGN.md 3.1.  The split happens once per element while the tile is staged into LDS.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16& b1, __bf16& b2, __bf16& b3) {
  b1 = (__bf16)x;
  float r = x - (float)b1;
  b2 = (__bf16)r;
  r -= (float)b2;
  b3 = (__bf16)r;
}

template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm_bn_bf16x6_kernel(GemmParams p, int tiles_m, int tiles_n) {
  drop_resolve(p.drop_in);
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDB = BK + 8;                         // 80 / 144-byte rows: conflict-free 16-byte fragment reads
  constexpr int C4 = BK / 4, A4 = BM * C4 / 256, B4 = BN * C4 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);       // [3][BM][LDB]
  __bf16* Bs = As + 3 * BM * LDB;                     // [3][BN][LDB]
  const int kc = p.K / p.split_k;
  const int k_base = blockIdx.y * kc;
  float* s_in = reinterpret_cast<float*>(Bs + 3 * BN * LDB);
  float* t_in = s_in + kc;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BM;
  const int n0 = tn_idx * BN;
  const bool act = p.stats_in != nullptr;
  if (act) {
    for (int kk = threadIdx.x; kk < kc; kk += 256)
      bn_affine(p.stats_in[k_base + kk], p.stats_in[p.K + k_base + kk], p.count, p.gamma_in[k_base + kk],
                p.beta_in[k_base + kk], s_in[kk], t_in[kk]);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  float4 ra[A4], rb[B4];
  auto load_tiles = [&](int kt) {
    const int k0 = k_base + kt * BK;
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      const int64_t row = m0 + r;
      ra[i] = row < p.M ? *reinterpret_cast<const float4*>(p.A + row * p.lda + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      const int n = n0 + r;
      rb[i] = n < p.Nout ? *reinterpret_cast<const float4*>(p.W + (int64_t)n * p.K + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put = [&](__bf16* base, int rows, int r, int c4, float4 v) {
    bf16x4 q1, q2, q3;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { __bf16 a, b, c; split3(x[j], a, b, c); q1[j] = a; q2[j] = b; q3[j] = c; }
    *reinterpret_cast<bf16x4*>(base + (0 * rows + r) * LDB + c4 * 4) = q1;
    *reinterpret_cast<bf16x4*>(base + (1 * rows + r) * LDB + c4 * 4) = q2;
    *reinterpret_cast<bf16x4*>(base + (2 * rows + r) * LDB + c4 * 4) = q3;
  };
  auto store_tiles = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      float4 v = ra[i];
      if (act) {
        const float4 s = *reinterpret_cast<const float4*>(s_in + k0 + c4 * 4);
        const float4 t = *reinterpret_cast<const float4*>(t_in + k0 + c4 * 4);
        v.x = fmaxf(fmaf(v.x, s.x, t.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s.y, t.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s.z, t.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s.w, t.w), 0.f);
        if (p.drop_in.on) {
          const unsigned long long idx = (unsigned long long)(m0 + r) * p.K + k_base + k0 + c4 * 4;
          float dv[4] = {v.x, v.y, v.z, v.w};
          drop_apply4(p.drop_in, p.drop_stream, idx, dv);        // (idx % 4 == 0: one hash for the four)
          v = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
      }
      put(As, BM, r, c4, v);
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      put(Bs, BN, r, c4, rb[i]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = kc / BK;
  load_tiles(0);
  const int a_off = (wm * TM * 32 + (lane & 31)) * LDB + (lane >> 5) * 8;
  const int b_off = (wn * TN * 32 + (lane & 31)) * LDB + (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tiles(kt);
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[TM][3], b[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(As + q * BM * LDB + a_off + i * 32 * LDB + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8*>(Bs + q * BN * LDB + b_off + j * 32 * LDB + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first so they are not absorbed one by one into a large accumulator
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }

  if (p.split_k > 1) {
    float* slab = p.slab + (size_t)blockIdx.y * p.M * p.Nout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.Nout) slab[row * p.Nout + col] = acc[i][j][r];
        }
    }
    return;
  }
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = wn * TN * 32 + j * 32 + (lane & 31);
    const int col = n0 + cl;
    const float bias = col < p.Nout ? p.bias[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M && col < p.Nout) {
          const float y = acc[i][j][r] + bias;
          p.Y[row * p.ldy + col] = y;
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
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += 256) {
    const int which = i / BN, cl = i % BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BN + cl] + colred[(1 * 2 + which) * BN + cl]);
  }
}

// ------------------------------------------------------------------------------------------------
// The same layer on the fp16 matrix cores at fp32 accuracy with HALF the bf16x6 work: x = h1 + h2 in fp16 carries
// 22 mantissa bits once the operand is scaled into fp16's normal range (exact power-of-two scales from absmax
// values gathered on the way: mtmc::amax_kernel for x and the weights, the producing layer's epilogue for Y), the
// 11x11-bit products are exact in the fp32 accumulator, and three of them (a1w1 + a1w2 + a2w1) leave a representation
// error of <= 1e-7 |a||w| -- below the rounding error of a plain fp32 dot product of this length.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kAffChunk = 512;   // input-affine columns resident in LDS (keeps two 128x128x64 blocks per CU)

template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm_bn_f16x3_kernel(GemmParams p, int tiles_m, int tiles_n) {
  drop_resolve(p.drop_in);
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDB = BK + 8;                         // 80 / 144-byte rows: conflict-free 16-byte fragment reads
  constexpr int C4 = BK / 4, A4 = BM * C4 / 256, B4 = BN * C4 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PSA = BM * LDB, PSB = BN * LDB;       // piece strides
  _Float16* As = reinterpret_cast<_Float16*>(smem);   // [2][PSA]
  _Float16* Bs = As + 2 * PSA;                        // [2][PSB]
  if (p.pass_blocks > 0 && (int)blockIdx.x >= (int)gridDim.x - p.pass_blocks) {   // passenger workgroups: enc2 (kernels.h)
    if (blockIdx.y == 0)
      enc2_body(p.pass_enc, p.pass_attr, p.pass_edges, p.pass_e_total, p.pass_stat,
                (int)blockIdx.x - ((int)gridDim.x - p.pass_blocks), p.pass_blocks);
    return;
  }
  const int kc = p.K / p.split_k;
  const int k_base = blockIdx.y * kc;
  float* s_in = reinterpret_cast<float*>(Bs + 2 * PSB);
  const int aff_n = kc < kAffChunk ? kc : kAffChunk;  // BatchNorm affine of the input: kAffChunk columns at a time
  float* t_in = s_in + aff_n;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = (slot / tiles_n) * 8 + xcd, tn_idx = slot % tiles_n;
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * BM;
  const int n0 = tn_idx * BN;
  const bool act = p.stats_in != nullptr;
  // Power-of-two scales that put the largest |operand| at 2^14 (fp16: 2^15 max, full 22-bit two-piece precision
  // down to 2^-3): exact to apply and to undo.  W: the layer's absmax; A: x's absmax (layer 0), or a bound on
  // relu(bn(Y_prev)) from Y_prev's absmax and this block's K-slice of the BatchNorm affine.
  float* sc = t_in + aff_n;                             // [4] scale of A, scale of W, 1 / (both)
  float* wred = sc + 4;                                 // [8]
  float ms = 0.f, mt = 0.f;
  auto fill_affine = [&](int chunk) {                   // s_in/t_in <- columns [chunk*kAffChunk, +aff_n) of the slice
    for (int kk = threadIdx.x; kk < aff_n; kk += 256) {
      const int kq = k_base + chunk * kAffChunk + kk;
      if (chunk * kAffChunk + kk < kc)
        bn_affine(p.stats_in[kq], p.stats_in[p.K + kq], p.count, p.gamma_in[kq], p.beta_in[kq], s_in[kk], t_in[kk]);
    }
  };
  if (act) {
    for (int kk = threadIdx.x; kk < kc; kk += 256) {     // all of the slice once, for the bound on |relu(bn(.))|
      float sv, tv;
      bn_affine(p.stats_in[k_base + kk], p.stats_in[p.K + k_base + kk], p.count, p.gamma_in[k_base + kk],
                p.beta_in[k_base + kk], sv, tv);
      if (kk < aff_n) { s_in[kk] = sv; t_in[kk] = tv; }
      ms = fmaxf(ms, fabsf(sv));
      mt = fmaxf(mt, fabsf(tv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ms = fmaxf(ms, __shfl_xor(ms, off, 64));
      mt = fmaxf(mt, __shfl_xor(mt, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6] = ms; wred[4 + (threadIdx.x >> 6)] = mt; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned ua = 0, uw = 0;                              // kAmaxRep replicas each (atomicMax chains kept short)
#pragma unroll
    for (int r = 0; r < kAmaxRep; ++r) { ua = max(ua, p.amax_a[r]); uw = max(uw, p.amax_w[r]); }
    float bound = __uint_as_float(ua);
    if (act) {
      const float s4 = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
      const float t4 = fmaxf(fmaxf(wred[4], wred[5]), fmaxf(wred[6], wred[7]));
      bound = fmaf(bound, s4, t4) * (p.drop_in.on ? p.drop_in.inv_keep : 1.f);
    }
    int ea = 0, ew = 0;
    const float aw = __uint_as_float(uw);
    if (bound > 0.f && bound < 3e38f) (void)frexpf(bound, &ea);
    if (aw > 0.f && aw < 3e38f) (void)frexpf(aw, &ew);
    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);        // keep every factor (and their product's two halves) finite
    ew = ew < -100 ? -100 : (ew > 100 ? 100 : ew);
    sc[0] = ldexpf(1.f, 14 - ea);
    sc[1] = ldexpf(1.f, 14 - ew);
    sc[2] = ldexpf(1.f, ea - 14);                         // undone in two steps: the product could leave fp32's range
    sc[3] = ldexpf(1.f, ew - 14);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  float4 ra[A4], rb[B4];
  // Rows past M / columns past Nout are loaded from the last valid row instead of being zero-filled: they only feed
  // accumulators the epilogue never stores, and unconditional loads keep the k-loop free of exec-mask branches.
  const float* a_src[A4];
  const float* b_src[B4];
#pragma unroll
  for (int i = 0; i < A4; ++i) {
    const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
    const int64_t row = m0 + r < p.M ? m0 + r : p.M - 1;
    a_src[i] = p.A + row * p.lda + k_base + c4 * 4;
  }
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
    const int n = n0 + r < p.Nout ? n0 + r : p.Nout - 1;
    b_src[i] = p.W + (int64_t)n * p.K + k_base + c4 * 4;
  }
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const float4*>(a_src[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const float4*>(b_src[i] + kt * BK);
  };
  auto put = [&](_Float16* base, int piece_stride, int r, int c4, float4 v, float scale) {
    typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
    const float x0 = v.x * scale, x1 = v.y * scale, x2 = v.z * scale, x3 = v.w * scale;
    // two-piece split with truncating conversions (v_cvt_pkrtz_f16_f32): h1 = rtz(x), h2 = rtz(x - h1); the
    // residual x - h1 is exact, 22 mantissa bits are kept.
    // Round 1 saw wrong results with the `(_Float16)x` spelling of this split and blamed the conversion instructions.
    // Root cause (round 2; stand-alone reproducer tools/hazard/pk_probe.hip): for that spelling hipcc's
    // SLP vectoriser multiplies .z/.w by the scale with `v_pk_mul_f32 / v_pk_fma_f32 D, A, S op_sel:[0,1(,0)]` (the scale
    // sits in the ODD register of the pair read from sc[]), and on gfx950 a packed-fp32 op whose LOW lane selects the
    // HIGH dword of a source returns 0 in that lane for lanes 48-63 while the wave's own MFMAs are still in flight and
    // a second wave shares the SIMD.  Not a property of this source line, so the guard is in the build: tools/check_isa.py
    // (run by the Makefile and by tests/test_isa_lint.py) rejects any MFMA kernel that contains the form.
#ifdef MTMC_F16_CVT_RNE   // the spelling that made hipcc emit the failing form (tools/hazard/ab_build.sh builds it on the side)
    {
      const float xs[4] = {x0, x1, x2, x3};
      f16x4 q1r, q2r;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const _Float16 h1 = (_Float16)xs[j]; q1r[j] = h1; q2r[j] = (_Float16)(xs[j] - (float)h1); }
      *reinterpret_cast<f16x4*>(base + r * LDB + c4 * 4) = q1r;
      *reinterpret_cast<f16x4*>(base + piece_stride + r * LDB + c4 * 4) = q2r;
      return;
    }
#endif
    const h2_t a01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), a23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    const h2_t b01 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)a01[0], x1 - (float)a01[1]);
    const h2_t b23 = __builtin_amdgcn_cvt_pkrtz(x2 - (float)a23[0], x3 - (float)a23[1]);
    uint2 q1, q2;
    q1.x = __builtin_bit_cast(unsigned, a01); q1.y = __builtin_bit_cast(unsigned, a23);
    q2.x = __builtin_bit_cast(unsigned, b01); q2.y = __builtin_bit_cast(unsigned, b23);
    *reinterpret_cast<uint2*>(base + r * LDB + c4 * 4) = q1;
    *reinterpret_cast<uint2*>(base + piece_stride + r * LDB + c4 * 4) = q2;
  };
  auto store_tiles = [&](int kt) {
    const int k0 = kt * BK;
    const float sa = sc[0], sw = sc[1];
#pragma unroll
    for (int i = 0; i < A4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      float4 v = ra[i];
      if (act) {
        const float4 s = *reinterpret_cast<const float4*>(s_in + (k0 % kAffChunk) + c4 * 4);
        const float4 t = *reinterpret_cast<const float4*>(t_in + (k0 % kAffChunk) + c4 * 4);
        v.x = fmaxf(fmaf(v.x, s.x, t.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s.y, t.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s.z, t.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s.w, t.w), 0.f);
        if (p.drop_in.on) {
          const unsigned long long idx = (unsigned long long)(m0 + r) * p.K + k_base + k0 + c4 * 4;
          float dv[4] = {v.x, v.y, v.z, v.w};
          drop_apply4(p.drop_in, p.drop_stream, idx, dv);        // (idx % 4 == 0: one hash for the four)
          v = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
      }
      put(As, PSA, r, c4, v, sa);
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      const int f = threadIdx.x + i * 256, r = f / C4, c4 = f % C4;
      put(Bs, PSB, r, c4, rb[i], sw);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = kc / BK;
  load_tiles(0);
  const int a_off = (wm * TM * 32 + (lane & 31)) * LDB + (lane >> 5) * 8;
  const int b_off = (wn * TN * 32 + (lane & 31)) * LDB + (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (act && kt > 0 && (kt * BK) % kAffChunk == 0) {   // next kAffChunk columns of the input affine
      fill_affine(kt * BK / kAffChunk);
      __syncthreads();
    }
    store_tiles(kt);
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 a[TM][2], b[TN][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) a[i][q] = *reinterpret_cast<const f16x8*>(As + q * PSA + a_off + i * 32 * LDB + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[j][q] = *reinterpret_cast<const f16x8*>(Bs + q * PSB + b_off + j * 32 * LDB + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // fp16 x fp16 products are exact in the fp32 accumulator; the dropped a2.w2 term is <= 2^-22 |a||w|
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }

  const float inv_a = sc[2], inv_w = sc[3];
  if (p.split_k > 1) {
    float* slab = p.slab + (size_t)blockIdx.y * p.M * p.Nout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.Nout) slab[row * p.Nout + col] = acc[i][j][r] * inv_a * inv_w;
        }
    }
    return;
  }
  __syncthreads();
  double* colred = reinterpret_cast<double*>(smem);
  float ymax = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = wn * TN * 32 + j * 32 + (lane & 31);
    const int col = n0 + cl;
    const float bias = col < p.Nout ? p.bias[col] : 0.f;
    double cs = 0, cq = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M && col < p.Nout) {
          const float y = fmaf(acc[i][j][r] * inv_a, inv_w, bias);
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
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += 256) {
    const int which = i / BN, cl = i % BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out)
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[(0 * 2 + which) * BN + cl] + colred[(1 * 2 + which) * BN + cl]);
  }
  if (p.amax_y) {                                        // the next layer scales its A operand from this
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    __syncthreads();                                     // colred consumed; reuse its first words
    float* wmax = reinterpret_cast<float*>(smem);
    if (lane == 0) wmax[wid] = ymax;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep),
                __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: remember it per (kernel, device), under a lock.
bool allow_big_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> g(mu);
  if (done.count({fn, dev})) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done.insert({fn, dev});
  return true;
}

template <int TM, int TN, int BK>
static int launch_f16x3(const GemmParams& p, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDB = BK + 8;
  const int tiles_m = (int)((p.M + BM - 1) / BM), tiles_n = (p.Nout + BN - 1) / BN;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const int kc = p.K / p.split_k;
  size_t lds = (size_t)2 * (BM + BN) * LDB * 2 + (size_t)(2 * (kc < kAffChunk ? kc : kAffChunk) + 12) * sizeof(float);
  const size_t epi = (size_t)4 * BN * sizeof(double);
  if (lds < epi) lds = epi;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_bn_f16x3_kernel<TM, TN, BK>), 128 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL((gemm_bn_f16x3_kernel<TM, TN, BK>), dim3(grid + p.pass_blocks, p.split_k), dim3(256), lds, s, p, tiles_m, tiles_n);
  return MTMC_OK;
}

template <int TM, int TN, int BK>
static int launch_bf16x6(const GemmParams& p, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDB = BK + 8;
  const int tiles_m = (int)((p.M + BM - 1) / BM), tiles_n = (p.Nout + BN - 1) / BN;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  size_t lds = (size_t)3 * (BM + BN) * LDB * 2 + (size_t)2 * (p.K / p.split_k) * sizeof(float);
  const size_t epi = (size_t)4 * BN * sizeof(double);
  if (lds < epi) lds = epi;
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_bn_bf16x6_kernel<TM, TN, BK>), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL((gemm_bn_bf16x6_kernel<TM, TN, BK>), dim3(grid, p.split_k), dim3(256), lds, s, p, tiles_m, tiles_n);
  return MTMC_OK;
}

// Second half of a split-K layer: Y = bias + sum of the K-slices' slabs (fixed order => reproducible),
// and the fp64 column statistics of Y.  Block = 64 columns x kCombRows rows; thread = one column, every 4th row;
// all of a row's slice loads are issued together (they are independent; only the additions are ordered).
constexpr int kCombRows = 16;
constexpr int kMaxSplit = 4;   // 8 measured slower with the fp16 kernel (S02 165.7 vs 161.4 us: half the slab traffic)

__global__ __launch_bounds__(256) void combine_stats_kernel(GemmParams p) {
  __shared__ double red[2 * 4 * 64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * kCombRows;
  const size_t slice = (size_t)p.M * p.Nout;
  double cs = 0, cq = 0;
  float ymax = 0.f;
  if (col < p.Nout) {
    const float bias = p.bias[col];
#pragma unroll
    for (int i = 0; i < kCombRows / 4; ++i) {
      const int64_t row = r0 + rg + 4 * i;
      if (row < p.M) {
        float v[kMaxSplit];
#pragma unroll
        for (int z = 0; z < kMaxSplit; ++z) v[z] = z < p.split_k ? p.slab[z * slice + row * p.Nout + col] : 0.f;
        float y = bias;
#pragma unroll
        for (int z = 0; z < kMaxSplit; ++z) y += v[z];
        p.Y[row * p.ldy + col] = y;
        ymax = fmaxf(ymax, fabsf(y));
        cs += y;
        cq += (double)y * y;
      }
    }
  }
  __shared__ float wmax[4];
  if (p.amax_y) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
    if (cl == 0) wmax[rg] = ymax;
  }
  red[rg * 64 + cl] = cs;
  red[256 + rg * 64 + cl] = cq;
  __syncthreads();
  if (p.amax_y && threadIdx.x == 0)
    atomicMax(p.amax_y + ((blockIdx.x + blockIdx.y) % kAmaxRep),
              __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  if (threadIdx.x < 128 && p.stats_out) {
    const int which = threadIdx.x >> 6, c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < p.Nout) {
      const double* q = red + which * 256 + (threadIdx.x & 63);
      unsafeAtomicAdd(p.stats_out + which * p.Nout + c, q[0] + q[64] + q[128] + q[192]);
    }
  }
}

// Generic path for shapes the MFMA kernel does not take (K not a multiple of 32, unaligned rows): the small
// edge-side layers when MLP.forward is used as a stand-alone op.  One thread per output element.
__global__ __launch_bounds__(256) void linear_generic_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* cs = reinterpret_cast<double*>(smem);       // [2][Nout]
  for (int i = threadIdx.x; i < 2 * p.Nout; i += 256) cs[i] = 0.0;
  __syncthreads();
  const int64_t total = p.M * p.Nout, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / p.Nout;
    const int c = (int)(i % p.Nout);
    float acc = p.bias[c];
    for (int k = 0; k < p.K; ++k) acc = fmaf(p.A[r * p.lda + k], p.W[(int64_t)c * p.K + k], acc);
    p.Y[r * p.ldy + c] = acc;
    if (p.stats_out) {
      atomicAdd(&cs[c], (double)acc);
      atomicAdd(&cs[p.Nout + c], (double)acc * acc);
    }
  }
  __syncthreads();
  if (p.stats_out)
    for (int i = threadIdx.x; i < 2 * p.Nout; i += 256) unsafeAtomicAdd(p.stats_out + i, cs[i]);
}

template <int TM, int TN, int BK>
static void launch_cfg(const GemmParams& p, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDK = BK + 4;
  const int tiles_m = (int)((p.M + BM - 1) / BM), tiles_n = (p.Nout + BN - 1) / BN;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  size_t lds = (size_t)(BM + BN) * LDK * sizeof(float) + (size_t)2 * (p.K / p.split_k) * sizeof(float);
  const size_t epi = (size_t)4 * BN * sizeof(double);
  if (lds < epi) lds = epi;
  hipLaunchKernelGGL((gemm_bn_kernel<TM, TN, BK>), dim3(grid, p.split_k), dim3(256), lds, s, p, tiles_m, tiles_n);
}

void launch_combine(const GemmParams& p, hipStream_t s) {
  hipLaunchKernelGGL(combine_stats_kernel, dim3((p.Nout + 63) / 64, (unsigned)((p.M + kCombRows - 1) / kCombRows)),
                     dim3(256), 0, s, p);
}

// 0: MFMA kernel not applicable; 1: 64x64 tiles; 2: 128x128 tiles.  *split_k > 1 only for few-row problems,
// where one block per output tile would leave most of the 256 CUs idle and serialise the whole K loop.
// K > 6144 is only a limit for layers with an input BatchNorm under the fp32 / bf16 kernels (their whole input affine
// sits in LDS); the fp16 kernel chunks it, and a product without an input affine (the weight-gradient GEMM of the
// backward reduces over the node rows: K = N) has none.
int gemm_plan(int64_t M, int K, int Nout, int* split_k, bool long_k_ok) {
  constexpr int BK = 64;
  *split_k = 1;
  if (K % 32 != 0 || (K > 6144 && !long_k_ok)) return 0;
  const int64_t big_tiles = ((M + 127) / 128) * ((Nout + 127) / 128);
  if (big_tiles >= 256 && Nout >= 128) return 2;   // one 128x128 tile per CU or more (fp16 kernel: 256 beats 512 at M = 6000)
  const int64_t tiles = ((M + 63) / 64) * ((Nout + 63) / 64);
  int s = 1;
  while (s < kMaxSplit && tiles * (s * 2) <= 1024 && K % (s * 2 * BK) == 0 && K / (s * 2) >= 64) s *= 2;
  // K <= 128: two k-tiles at most -- a slice per block saves nothing and costs the combine launch (measured, S02)
  if (tiles < 256 && K > 128) *split_k = s;
  return 1;
}

// which: 1 = the GEMM only, 2 = the split-K combine only (no-op for un-split layers), 3 = both
int launch_gemm_bn(const GemmParams& p, hipStream_t s, int which) {
  if (p.M < 1 || p.Nout < 1 || p.K < 1) return MTMC_E_ARG;
  const bool fp32_only = knobs().gemm_fp32, no_f16 = knobs().gemm_no_f16;
  const bool f16_ok = p.amax_a && p.amax_w && !fp32_only && !no_f16;
  const bool long_k_ok = f16_ok;          // the fp16 kernel keeps at most kAffChunk affine columns in LDS at a time
  if (p.K % 32 != 0 || (p.K > 6144 && !long_k_ok) || (p.lda % 4) != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) {
    if (p.stats_in != nullptr || p.Nout > 2048) return MTMC_E_ARG;
    if (!(which & 1)) return MTMC_OK;
    const int64_t blocks = (p.M * p.Nout + 255) / 256;
    hipLaunchKernelGGL(linear_generic_kernel, dim3((int)(blocks > 2048 ? 2048 : blocks)), dim3(256),
                       (size_t)2 * p.Nout * sizeof(double), s, p);
    if (p.pass_blocks > 0) launch_enc2(p.pass_enc, p.pass_attr, p.pass_edges, p.pass_e_total, p.pass_stat, s);
    return MTMC_OK;
  }
  GemmParams q = p;
  int sk;
  const int cfg = gemm_plan(p.M, p.K, p.Nout, &sk, long_k_ok);
  q.split_k = (p.slab != nullptr) ? sk : 1;
  // the passenger job (enc2) rides in the 64 x 64-tile fp16 kernel's launch; any other kernel: its own launch behind the GEMM
  const bool carries = cfg == 1 && f16_ok && q.split_k == 1;
  const bool job_behind = p.pass_blocks > 0 && !carries;
  if (!carries) q.pass_blocks = 0;
  if (which & 1) {
    const bool f16 = f16_ok;
    int rc = MTMC_OK;
    if (cfg == 2 && f16 && p.K % 64 == 0) rc = launch_f16x3<2, 2, 64>(q, s);
    else if (cfg == 2 && f16) rc = launch_f16x3<2, 2, 32>(q, s);
    else if (cfg == 1 && f16 && (p.K / q.split_k) % 64 == 0) rc = launch_f16x3<1, 1, 64>(q, s);
    else if (cfg == 1 && f16) rc = launch_f16x3<1, 1, 32>(q, s);
    else if (cfg == 2 && !fp32_only) rc = launch_bf16x6<2, 2, 32>(q, s);
    else if (cfg == 2) launch_cfg<2, 2, 32>(q, s);
    else if ((p.K / q.split_k) % 64 == 0) launch_cfg<1, 1, 64>(q, s);
    else launch_cfg<1, 1, 32>(q, s);
    if (rc != MTMC_OK) return rc;
    if (job_behind) launch_enc2(p.pass_enc, p.pass_attr, p.pass_edges, p.pass_e_total, p.pass_stat, s);
  }
  if ((which & 2) && q.split_k > 1) launch_combine(q, s);
  return MTMC_OK;
}

}  // namespace mtmc
