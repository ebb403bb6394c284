// The last, narrow node-encoder layers of MANY-ROW graphs (reference models/mlp.py:14-27 via models/mpn.py:131; default
// config: 128 -> 32): Y = relu(bn(Y_prev)) . W^T + b as a ROW-STREAMING kernel -- one wave per 16 rows, the whole weight
// matrix in registers, no LDS staging and no barrier in the loop.
//
// A tiled GEMM spends such a layer in prologues and epilogues (1563 tiles of 64 x 64 with two k-tiles each: 48 us at
// 100000 rows for 64 MB of compulsory traffic).  Here the layer is what it is, a pass over [M][K] at HBM speed:
//   * W [N][K] is split into its two fp16 pieces ONCE per wave, with one power-of-two scale per row, straight into the B
//     operands of v_mfma_f32_16x16x32_f16 (lane (n = l % 16, g = l / 16) holds W[n][32 s + 8 g .. + 7] of k-step s):
//     (K / 32) x (N / 16) x 2 fragments of 4 registers -- 64 registers for 128 -> 32;
//   * per 16 rows a lane loads row l % 16, floats 32 s + 8 g .. + 7 of every k-step (four lanes cover a 128-byte line),
//     applies the previous layer's BatchNorm + ReLU and the operand scale, splits (v_cvt_pkrtz) and feeds the MFMAs
//     directly: three products per fp32 product, as in the other encoder kernels;
//   * bias, raw Y, fp64 column statistics (kept per lane over all of a wave's rows, reduced once at the end), |Y|max.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "lds_dma.h"

namespace mtmc {

template <int KS, int NB>    // K = 32 KS, Nout = 16 NB
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmParams p) {
  constexpr int K = 32 * KS, N = 16 * NB;
  __shared__ float s_in[K], t_in[K];
  __shared__ float wred[8];
  __shared__ float sc[2];
  __shared__ double colred[4][2][N];
  __shared__ float wmax[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;

  // ---- prologue: input BatchNorm affine, the bound on |relu(bn(.))| -> one power-of-two scale for A
  {
    float ms = 0.f, mt = 0.f;
    for (int kk = threadIdx.x; kk < K; kk += 256) {
      float sv, tv;
      bn_affine(p.stats_in[kk], p.stats_in[K + kk], p.count, p.gamma_in[kk], p.beta_in[kk], sv, tv);
      s_in[kk] = sv; t_in[kk] = tv;
      ms = fmaxf(ms, fabsf(sv));
      mt = fmaxf(mt, fabsf(tv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ms = fmaxf(ms, __shfl_xor(ms, off, 64));
      mt = fmaxf(mt, __shfl_xor(mt, off, 64));
    }
    if (lane == 0) { wred[wid] = ms; wred[4 + wid] = mt; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned ua = 0;
#pragma unroll
    for (int r = 0; r < kAmaxRep; ++r) ua = max(ua, p.amax_a[r]);
    const float s4 = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    const float t4 = fmaxf(fmaxf(wred[4], wred[5]), fmaxf(wred[6], wred[7]));
    const float bound = fmaf(__uint_as_float(ua), s4, t4);
    int ea = 0;
    if (bound > 0.f && bound < 3e38f) (void)frexpf(bound, &ea);
    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
    sc[0] = ldexpf(1.f, 14 - ea);
    sc[1] = ldexpf(1.f, ea - 14);
  }
  __syncthreads();
  const float sa = sc[0], inv_a = sc[1];

  // ---- W -> B fragments in registers (per wave), one scale per output row n
  f16x8 bw[NB][KS][2];
  float inv_w[NB], bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = 16 * nb + i16;
    float wv[KS][8];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* src = p.W + (int64_t)n * K + 32 * s + 8 * g;
      const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
      wv[s][0] = v0.x; wv[s][1] = v0.y; wv[s][2] = v0.z; wv[s][3] = v0.w;
      wv[s][4] = v1.x; wv[s][5] = v1.y; wv[s][6] = v1.z; wv[s][7] = v1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(wv[s][j]));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));             // the four lanes that share row n
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    int e = 0;
    if (m > 0.f && m < 3e38f) (void)frexpf(m, &e);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sw = ldexpf(1.f, 14 - e);
    inv_w[nb] = ldexpf(1.f, e - 14);
    bias[nb] = p.bias[n];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      h2_t hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = wv[s][2 * j] * sw, x1 = wv[s][2 * j + 1] * sw;
        hi[j] = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        lo[j] = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi[j][0], x1 - (float)hi[j][1]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bw[nb][s][0][2 * j] = (_Float16)hi[j][0]; bw[nb][s][0][2 * j + 1] = (_Float16)hi[j][1];
        bw[nb][s][1][2 * j] = (_Float16)lo[j][0]; bw[nb][s][1][2 * j + 1] = (_Float16)lo[j][1];
      }
    }
  }

  double cs[NB], cq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) cs[nb] = cq[nb] = 0;
  float ymax = 0.f;
  const int64_t n_groups = (p.M + 15) / 16;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wid; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t row = grp * 16 + i16;
    const float* src = p.A + (row < p.M ? row : p.M - 1) * p.lda + 8 * g;     // rows past M: a valid row (never stored)
    float4 v[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      v[s][0] = *reinterpret_cast<const float4*>(src + 32 * s);
      v[s][1] = *reinterpret_cast<const float4*>(src + 32 * s + 4);
    }
    f32x4v acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 s0 = *reinterpret_cast<const float4*>(s_in + 32 * s + 8 * g), s1 = *reinterpret_cast<const float4*>(s_in + 32 * s + 8 * g + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(t_in + 32 * s + 8 * g), t1 = *reinterpret_cast<const float4*>(t_in + 32 * s + 8 * g + 4);
      const float x[8] = {v[s][0].x, v[s][0].y, v[s][0].z, v[s][0].w, v[s][1].x, v[s][1].y, v[s][1].z, v[s][1].w};
      const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      f16x8 a1, a2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xa = fmaxf(fmaf(x[2 * j], sv[2 * j], tv[2 * j]), 0.f) * sa;
        const float xb = fmaxf(fmaf(x[2 * j + 1], sv[2 * j + 1], tv[2 * j + 1]), 0.f) * sa;
        const h2_t hi = __builtin_amdgcn_cvt_pkrtz(xa, xb);
        const h2_t lo = __builtin_amdgcn_cvt_pkrtz(xa - (float)hi[0], xb - (float)hi[1]);
        a1[2 * j] = (_Float16)hi[0]; a1[2 * j + 1] = (_Float16)hi[1];
        a2[2 * j] = (_Float16)lo[0]; a2[2 * j + 1] = (_Float16)lo[1];
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bw[nb][s][0], acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bw[nb][s][1], acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bw[nb][s][0], acc[nb], 0, 0, 0);
      }
    }
    // a lane holds rows 4 g + r of column i16 of every block
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = grp * 16 + 4 * g + r;
        if (orow < p.M) {
          const float y = fmaf(acc[nb][r] * inv_a, inv_w[nb], bias[nb]);
          p.Y[orow * p.ldy + 16 * nb + i16] = y;
          ymax = fmaxf(ymax, fabsf(y));
          cs[nb] += y;
          cq[nb] += (double)y * y;
        }
      }
  }

  // ---- column statistics: lanes g = 0..3 share a column; then the four waves; one atomic per column and block
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    double a = cs[nb], b = cq[nb];
    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
    if (lane < 16) { colred[wid][0][16 * nb + i16] = a; colred[wid][1][16 * nb + i16] = b; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
  if (lane == 0) wmax[wid] = ymax;
  __syncthreads();
  if (p.stats_out)
    for (int i = threadIdx.x; i < 2 * N; i += 256) {
      const int which = i / N, col = i % N;
      unsafeAtomicAdd(p.stats_out + which * N + col,
                      colred[0][which][col] + colred[1][which][col] + colred[2][which][col] + colred[3][which][col]);
    }
  if (p.amax_y && threadIdx.x == 0)
    atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// which layers: eval mode, an input BatchNorm, many rows, and a weight matrix that fits the register file as MFMA operands
bool rows_layer(int64_t rows, int K, int Nout) {
  const Knobs& kn = knobs();
  if (kn.gemm_no_staged || kn.gemm_fp32 || kn.gemm_no_f16) return false;
  return rows >= 4096 && K == 128 && Nout == 32;
}

int launch_gemm_rows(const GemmParams& p, hipStream_t s) {
  if (p.M < 1 || p.K != 128 || p.Nout != 32 || !p.stats_in || !p.amax_a || p.drop_in.on || (p.lda & 3) || ((uintptr_t)p.A & 15) ||
      ((uintptr_t)p.W & 15))
    return 1;
  const int64_t groups = (p.M + 15) / 16, blocks = (groups + 3) / 4;
  // ~8 groups per wave: the per-wave prologue (weights -> fragments) is paid once per wave
  const int64_t want = (blocks + 7) / 8;
  const int grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL((gemm_rows_kernel<4, 2>), dim3(grid), dim3(256), 0, s, p);
  return MTMC_OK;
}

}  // namespace mtmc
