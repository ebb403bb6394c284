// Per-node kernels of the message-passing rounds (tiny next to the edge passes: O(N*32*40) flops).
//
//   node_proj_kernel  Pr|Pc = [h0|h] . We[:, node cols]^T  (4+4 per node),  Q = [h0|h] . Wn[:, node cols]^T (32)
//                     -- the algebraic form of the reference's x[row], x[col] gathers + cat + Linear
//                        (reference models/mpn.py:48,68 and :97-98): gather 16 B per edge end instead of 128 B
//                     + the node-only part of the node-update (z2) BatchNorm statistics, sum_i deg_i qb_ik and
//                     sum_i deg_i qb_ik^2, on few-edge lists (pass B adds the edge-dependent part: round 4)
//   node_stat_kernel  many-edge lists: sum / sum of squares over all E edges of z2 = Q[row] + A.e' + b, from per-node
//                     segment sums of e' and the degree (no pass over the edges)
//   bn_relu_rows      h0 = relu(bn(Y_last))
//   h_final           latent_node_feats output (mean aggregation divides by max(deg,1))
#include "kernels.h"
#include "rows_body.h"

namespace mtmc {

#ifndef MTMC_PROJ_MFMA_MIN
#define MTMC_PROJ_MFMA_MIN 1     // (rounds 3-4: 4096 -- few-node graphs kept the LDS kernel, "both at the launch floor"; stamped in
#endif                           //  round 5, the LDS kernel is 4.2 us of dependent round trips per workgroup, this one 2: DESIGN.md 5)
constexpr int64_t kProjMfmaMinRows = MTMC_PROJ_MFMA_MIN;   // node rows from which the matrix-core projection kernel is used
constexpr int kProjNodes = 16;   // nodes per block iteration (16 lanes per node: 8 output groups x 2 halves of the k range)
constexpr int kProjOut = 40;     // 4 (Pr) + 4 (Pc) + 32 (Q)

__global__ __launch_bounds__(256) void node_proj_kernel(NodeProjParams p) {
  drop_resolve(p.drop);
  __shared__ float wt[64 * kProjOut];              // [k][j], k < hn
  __shared__ float hs[kProjNodes * 66];            // [node][k], row stride hn+2 (see the k loop)
  __shared__ float ys[kH], yt[kH];                 // BatchNorm affine of the encoder's last layer (fused round 0)
  __shared__ EdgeEncAffine enc_af;
  __shared__ double scratch[kStatAttr + kStatEnc2];
  __shared__ double zred[4][2 * kH];               // node-only part of the z2 statistics per wave: sum deg qb | sum deg qb^2
  double zs1[4] = {0, 0, 0, 0}, zs2[4] = {0, 0, 0, 0};
  const int hn = p.hn, ldh = hn + 2;
  EK_T(0, 0);
  for (int i = threadIdx.x; i < hn * kProjOut; i += blockDim.x) {
    const int kk = i / kProjOut, j = i % kProjOut;
    float w;
    if (j < 4) w = p.ue_w[j * p.ue_ld + kk];
    else if (j < 8) w = p.ue_w[(j - 4) * p.ue_ld + hn + kk];
    else w = p.un_w[(j - 8) * p.un_ld + kk];
    wt[i] = w;
  }
  if (p.y_last && threadIdx.x < kH)
    bn_affine(p.y_stats[threadIdx.x], p.y_stats[kH + threadIdx.x], p.y_count, p.y_gamma[threadIdx.x],
              p.y_beta[threadIdx.x], ys[threadIdx.x], yt[threadIdx.x]);
  const int nl = threadIdx.x >> 4, part = (threadIdx.x >> 1) & 7, kh = threadIdx.x & 1;
  float ubv[4] = {0.f, 0.f, 0.f, 0.f};             // un_b of this lane's four Q channels (node-only z2 statistics)
  if (p.z2_stats) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ubv[i] = p.un_b[part + 8 * i];
  }
  const int64_t n_groups = (p.node_end - p.node_begin + kProjNodes - 1) / kProjNodes;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t node0 = p.node_begin + g * kProjNodes;
    // (the node's out-degree: requested here, used after the projections)
    const double dnode = (p.z2_stats && node0 + nl < p.node_end) ? (double)p.edge_deg[node0 + nl] : 0.0;
    __syncthreads();                               // wt/ys ready / previous hs consumed
    EK_T(0, 1);
    // stage the h rows of 32 nodes (1024 floats), then mirror / fetch the h0 half when reattaching
    for (int i = threadIdx.x; i < kProjNodes * kH; i += blockDim.x) {
      const int n = i >> 5, kk = i & 31;
      const int64_t node = node0 + n;
      float v = 0.f;
      if (node < p.node_end) {
        if (p.y_last) {
          v = drop_apply(p.drop, p.drop_stream, (unsigned long long)node * kH + kk,
                         fmaxf(fmaf(p.y_last[node * kH + kk], ys[kk], yt[kk]), 0.f));
          p.h0_out[node * kH + kk] = v;
        } else {
          v = p.h_src[node * kH + kk];
          if (p.deg) { const int d = p.deg[node]; v = v / (float)(d > 1 ? d : 1); }
        }
      }
      hs[n * ldh + (hn - kH) + kk] = v;
      if (hn == 2 * kH) hs[n * ldh + kk] = (p.y_last || node >= p.node_end) ? v : p.h0[node * kH + kk];
    }
    if (p.zero_buf && threadIdx.x < kProjNodes * 8) {   // 16 nodes x 32 floats = 128 float4
      const int64_t node = node0 + (threadIdx.x >> 3);
      if (node < p.node_end)
        reinterpret_cast<float4*>(p.zero_buf + node * kH)[threadIdx.x & 7] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    EK_T(0, 2);
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // a lane pair takes the even / the odd k: with row stride hn+2 the eight (node, k) words and the sixteen weights a
    // wave reads per step sit on different banks (halves of the k range, 32 apart, met on the same banks: 42 % of this
    // kernel's LDS cycles were conflicts at config 4)
    for (int kk = kh; kk < hn; kk += 2) {
      const float hv = hs[nl * ldh + kk];
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i] = fmaf(hv, wt[kk * kProjOut + part + 8 * i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] += __shfl_xor(acc[i], 1, 64);
    const int64_t node = node0 + nl;
    if (kh == 0 && node < p.node_end) {
      p.P[(part < 4 ? node : p.n_nodes + node) * 4 + (part & 3)] = acc[0];      // [Pr | Pc], see edge_z1
#pragma unroll
      for (int i = 1; i < 5; ++i) p.Q[node * kH + part + 8 * (i - 1)] = acc[i];
      if (p.z2_stats) {
#pragma unroll
        for (int i = 1; i < 5; ++i) {
          const double qb = (double)(acc[i] + ubv[i - 1]);
          zs1[i - 1] += dnode * qb;
          zs2[i - 1] += dnode * qb * qb;
        }
      }
    }
  }
  EK_T(0, 3);
  if (p.z2_stats) {        // a wave holds 4 node lanes (bits 4, 5 of the lane id) x 16: fold them, then the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      zs1[i] += __shfl_xor(zs1[i], 16, 64); zs2[i] += __shfl_xor(zs2[i], 16, 64);
      zs1[i] += __shfl_xor(zs1[i], 32, 64); zs2[i] += __shfl_xor(zs2[i], 32, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) < 16 && kh == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        zred[threadIdx.x >> 6][part + 8 * i] = zs1[i];
        zred[threadIdx.x >> 6][kH + part + 8 * i] = zs2[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kH)
      unsafeAtomicAdd(p.z2_stats + (blockIdx.x % kStatRep) * kZ2Stride + threadIdx.x,
                      zred[0][threadIdx.x] + zred[1][threadIdx.x] + zred[2][threadIdx.x] + zred[3][threadIdx.x]);
  }
  EK_T(0, 4);
  if (p.finalize_enc && blockIdx.x == 0) {         // once per forward: the edge encoder's two BatchNorm affines
    edge_enc_affine_to_smem(p.enc, p.e_total, 2, &enc_af, scratch);
    if (threadIdx.x < 16) p.enc.aff[threadIdx.x] = reinterpret_cast<const float*>(&enc_af)[threadIdx.x];
  }
  EK_T(0, 5);
}

// The same projections on the matrix cores, one wave per 16 nodes, no LDS and no barriers: [16 nodes][hn] . [hn][40 -> 48]
// as v_mfma_f32_16x16x4_f32 (exact fp32: 64 FLOP/clk/SIMD, 24 instructions per 16 nodes).  The K index is permuted so that
// the loads are whole rows: lane (i = l % 16, g = l / 16) holds h[node i][8g .. 8g+7] (two float4: four lanes cover a
// 128-byte row) and step s of the contraction takes k = 8g + s from every lane -- the weights are read with the same
// permutation once per wave and stay in registers.  Output block 0 / 1 = Q columns 0-15 / 16-31, block 2 = Pr | Pc | 0.
// Used for many-node graphs (a [N,32] pass at HBM speed: 24 -> 10 us per round at N = 100k); few-node graphs keep the
// kernel above (both sit at the launch floor there).
typedef float f32x4p __attribute__((ext_vector_type(4)));

template <int HALVES>   // 1: hn = 32 (h only); 2: hn = 64 ([h0 | h], reattach_initial_nodes)
__global__ __launch_bounds__(256) void node_proj_mfma_kernel(NodeProjParams p) {
  drop_resolve(p.drop);
  __shared__ float ys[kH], yt[kH];
  __shared__ EdgeEncAffine enc_af;
  __shared__ double scratch[kStatAttr + kStatEnc2];
  __shared__ double zred[4][2 * kH];               // node-only part of the z2 statistics per wave: sum deg qb | sum deg qb^2
  double zs1[2] = {0, 0}, zs2[2] = {0, 0};         // columns i16 and 16 + i16, this lane's rows
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const float ub0 = p.z2_stats ? p.un_b[i16] : 0.f, ub1 = p.z2_stats ? p.un_b[16 + i16] : 0.f;
  EK_T(5, 0);
  // B fragments: bw[nb][half][s] = W_out[n = 16 nb + i16][k = 32 half + 8 g + s].  Requested FIRST: the (fused round 0)
  // BatchNorm statistics of the encoder's last layer, their barrier and the first group's rows then travel beside them
  // instead of behind them (few-node graphs: the kernel is three dependent round trips otherwise)
  float bw[3][HALVES][8];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf)
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int kk = 32 * hf + 8 * g + s8;
        float w = 0.f;
        if (nb < 2) w = p.un_w[(16 * nb + i16) * p.un_ld + kk];
        else if (i16 < 4) w = p.ue_w[i16 * p.ue_ld + kk];
        else if (i16 < 8) w = p.ue_w[(i16 - 4) * p.ue_ld + p.hn + kk];
        bw[nb][hf][s8] = w;
      }
  if (p.y_last) {
    if (threadIdx.x < kH)
      bn_affine(p.y_stats[threadIdx.x], p.y_stats[kH + threadIdx.x], p.y_count, p.y_gamma[threadIdx.x],
                p.y_beta[threadIdx.x], ys[threadIdx.x], yt[threadIdx.x]);
    __syncthreads();
  }
  float ysv[8], ytv[8];
  if (p.y_last) {
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { ysv[s8] = ys[8 * g + s8]; ytv[s8] = yt[8 * g + s8]; }
  }
  EK_T(5, 1);
  const int64_t n_groups = (p.node_end - p.node_begin + 15) / 16;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wid; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t node = p.node_begin + grp * 16 + i16;
    const bool live = node < p.node_end;
    const int64_t nd = live ? node : p.node_end - 1;
    // (the out-degrees of this lane's four OUTPUT rows: requested with the rows, used behind the MFMAs)
    double dg[4] = {0, 0, 0, 0};
    if (p.z2_stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t on = p.node_begin + grp * 16 + 4 * g + r;
        dg[r] = (double)p.edge_deg[on < p.node_end ? on : p.node_end - 1];
      }
    }
    float a[HALVES][8];
    {   // the h half (the LAST 32 inputs when reattaching)
      const float* src = (p.y_last ? p.y_last : p.h_src) + nd * kH + 8 * g;
      const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (p.y_last) {
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
          v[s8] = drop_apply(p.drop, p.drop_stream, (unsigned long long)nd * kH + 8 * g + s8,
                             fmaxf(fmaf(v[s8], ysv[s8], ytv[s8]), 0.f));
        if (live) {
          *reinterpret_cast<float4*>(p.h0_out + node * kH + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(p.h0_out + node * kH + 8 * g + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      } else if (p.deg) {
        const int d = p.deg[nd];
        const float sc = (float)(d > 1 ? d : 1);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) v[s8] = v[s8] / sc;
      }
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) a[HALVES - 1][s8] = v[s8];
      if (HALVES == 2) {
        if (p.y_last) {
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) a[0][s8] = v[s8];          // first round: h0 == h
        } else {
          const float* s0 = p.h0 + nd * kH + 8 * g;
          const float4 u0 = *reinterpret_cast<const float4*>(s0), u1 = *reinterpret_cast<const float4*>(s0 + 4);
          a[0][0] = u0.x; a[0][1] = u0.y; a[0][2] = u0.z; a[0][3] = u0.w;
          a[0][4] = u1.x; a[0][5] = u1.y; a[0][6] = u1.z; a[0][7] = u1.w;
        }
      }
    }
    if (p.zero_buf && live) {
      *reinterpret_cast<float4*>(p.zero_buf + node * kH + 8 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p.zero_buf + node * kH + 8 * g + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4p acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      acc[nb] = f32x4p{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < HALVES; ++hf)
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[hf][s8], bw[nb][hf][s8], acc[nb], 0, 0, 0);
    }
    // a lane holds rows 4 g + r (r = 0..3) of column i16 of every block
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t on = p.node_begin + grp * 16 + 4 * g + r;
      if (on < p.node_end) {
        p.Q[on * kH + i16] = acc[0][r];
        p.Q[on * kH + 16 + i16] = acc[1][r];
        if (i16 < 8) p.P[(i16 < 4 ? on : p.n_nodes + on) * 4 + (i16 & 3)] = acc[2][r];
        if (p.z2_stats) {
          const double d = dg[r];
          const double q0 = (double)(acc[0][r] + ub0), q1 = (double)(acc[1][r] + ub1);
          zs1[0] += d * q0; zs2[0] += d * q0 * q0;
          zs1[1] += d * q1; zs2[1] += d * q1 * q1;
        }
      }
    }
  }
  EK_T(5, 2);
  if (p.z2_stats) {                                // fold the four row groups of a wave, then the waves through LDS
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      zs1[c] += __shfl_xor(zs1[c], 16, 64); zs2[c] += __shfl_xor(zs2[c], 16, 64);
      zs1[c] += __shfl_xor(zs1[c], 32, 64); zs2[c] += __shfl_xor(zs2[c], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        zred[wid][16 * c + i16] = zs1[c];
        zred[wid][kH + 16 * c + i16] = zs2[c];
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kH)
      unsafeAtomicAdd(p.z2_stats + (blockIdx.x % kStatRep) * kZ2Stride + threadIdx.x,
                      zred[0][threadIdx.x] + zred[1][threadIdx.x] + zred[2][threadIdx.x] + zred[3][threadIdx.x]);
  }
  EK_T(5, 3);
  if (p.finalize_enc && blockIdx.x == 0) {         // once per forward: the edge encoder's two BatchNorm affines
    __syncthreads();
    edge_enc_affine_to_smem(p.enc, p.e_total, 2, &enc_af, scratch);
    if (threadIdx.x < 16) p.enc.aff[threadIdx.x] = reinterpret_cast<const float*>(&enc_af)[threadIdx.x];
  }
  EK_T(5, 4);
}

__global__ __launch_bounds__(256) void node_stat_kernel(NodeStatParams p) {
  __shared__ double red[2 * 8 * 32];
  const int k = threadIdx.x & 31, slot = threadIdx.x >> 5;
  float a[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = p.un_w[k * p.un_ld + p.un_eoff + j];
  const float b = p.un_b[k];
  double s1 = 0, s2 = 0;
  const int64_t n_groups = (p.node_end - p.node_begin + 7) / 8;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t node = p.node_begin + g * 8 + slot;
    if (node < p.node_end) {
      const double d = (double)p.deg[node];
      const double qb = (double)(p.Q[node * kH + k] + b);
      double proj = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) proj += (double)a[j] * p.seg[node * 4 + j];
      s1 += d * qb + proj;
      s2 += d * qb * qb + 2.0 * qb * proj;
    }
    __syncthreads();                               // every channel has read seg[node]
    if (node < p.node_end && k < 4) p.seg[node * 4 + k] = 0.0;   // ready for the next round
  }
  red[slot * 32 + k] = s1;
  red[256 + slot * 32 + k] = s2;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int kk = threadIdx.x & 31, which = threadIdx.x >> 5;
    double s = 0;
    for (int sl = 0; sl < 8; ++sl) s += red[which * 256 + sl * 32 + kk];
    unsafeAtomicAdd(p.stats + kRoundZ2Off + (blockIdx.x % kStatRep) * kZ2Stride + which * 32 + kk, s);
  }
}

__global__ __launch_bounds__(256) void bn_relu_rows_kernel(const float* Y, int64_t ldy, int64_t rows, int dim,
                                                           const double* stats, const float* gamma, const float* beta,
                                                           double count, float* dst, Drop drop, unsigned drop_stream,
                                                           int64_t row0, unsigned* amax_out, float* dstT, int64_t ldt) {
  drop_resolve(drop);
  const int64_t total = rows * dim;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  float vmax = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthreads) {
    const int64_t r = i / dim;
    const int c = (int)(i % dim);
    float s, t;
    bn_affine(stats[c], stats[dim + c], count, gamma[c], beta[c], s, t);
    const float v = drop_apply(drop, drop_stream, (unsigned long long)(row0 + r) * dim + c, fmaxf(fmaf(Y[r * ldy + c], s, t), 0.f));
    dst[i] = v;
    if (dstT) dstT[(int64_t)c * ldt + r] = v;
    vmax = fmaxf(vmax, v);
  }
  if (dstT)                                          // padding rows of the transposed copy
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (ldt - rows) * dim; i += nthreads)
      dstT[(i % dim) * ldt + rows + i / dim] = 0.f;
  if (amax_out) {                                  // (backward: operand scale of the weight-gradient GEMM)
    __shared__ float wmax[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_publish(amax_out + blockIdx.x % kAmaxRep, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// The same with a TRANSPOSED copy beside it (the backward's recomputation of a layer's input activation, whose transpose is
// the weight-gradient GEMM's operand): 16 rows x 64 columns per workgroup, a thread keeps ONE column -- its BatchNorm affine is
// derived once, not per element -- and the transposed tile leaves through LDS as one 16-byte store per thread (round 5: the
// element-per-thread loop above took 6 / 10 / 15 us for 430 x 128 / 512 / 1024, most of it fp64 affines and 4-byte scattered stores).
__global__ __launch_bounds__(256) void bn_relu_rows_t_kernel(RowsTJob j) {
  drop_resolve(j.drop);
  bn_relu_rows_t_body(j, blockIdx.x, blockIdx.y, gridDim.y);
}

__global__ __launch_bounds__(256) void h_final_kernel(const float* src, const int* deg, int mean, int64_t n_nodes,
                                                      float* dst) {
  const int64_t total = n_nodes * kH;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthreads) {
    float v = src[i];
    if (mean) { const int d = deg[i / kH]; v = v / (float)(d > 1 ? d : 1); }
    dst[i] = v;
  }
}

static inline int cap_grid(int64_t blocks) { return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks)); }

void launch_node_proj(const NodeProjParams& p, hipStream_t s) {
  const int64_t rows = p.node_end - p.node_begin;
  if (rows >= kProjMfmaMinRows && (p.hn == kH || p.hn == 2 * kH)) {     // many nodes: one wave per 16 nodes on the matrix cores
    const int grid = cap_grid((rows + 63) / 64);
    if (p.hn == kH) hipLaunchKernelGGL(node_proj_mfma_kernel<1>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(node_proj_mfma_kernel<2>, dim3(grid), dim3(256), 0, s, p);
    return;
  }
  hipLaunchKernelGGL(node_proj_kernel, dim3(cap_grid((rows + kProjNodes - 1) / kProjNodes)), dim3(256), 0, s, p);
}
void launch_node_stat(const NodeStatParams& p, hipStream_t s) {
  hipLaunchKernelGGL(node_stat_kernel, dim3(cap_grid((p.node_end - p.node_begin + 7) / 8)), dim3(256), 0, s, p);
}
void launch_bn_relu_rows(const float* Y, int64_t ldy, int64_t rows, int dim, const double* stats, const float* gamma,
                         const float* beta, double count, float* dst, Drop drop, unsigned drop_stream, int64_t row0,
                         hipStream_t s, unsigned* amax_out, float* dstT, int64_t ldt) {
  if (dstT && ldt % 4 == 0 && rows > 0 && (rows + 15) / 16 <= 65535) {
    const RowsTJob j = {Y, ldy, rows, dim, stats, gamma, beta, count, dst, drop, drop_stream, row0, amax_out, dstT, ldt};
    hipLaunchKernelGGL(bn_relu_rows_t_kernel, dim3((dim + 63) / 64, (unsigned)((rows + 15) / 16)), dim3(256), 0, s, j);
    return;
  }
  // with the |.|max bookkeeping: one workgroup per CU, so that at most 16 of them meet on a word
  const int64_t blocks = (rows * dim + 255) / 256;
  hipLaunchKernelGGL(bn_relu_rows_kernel, dim3(amax_out && blocks > 256 ? 256 : cap_grid(blocks)), dim3(256), 0, s, Y, ldy,
                     rows, dim, stats, gamma, beta, count, dst, drop, drop_stream, row0, amax_out, dstT, ldt);
}
void launch_h_final(const float* src, const int* deg, int mean, int64_t n_nodes, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(h_final_kernel, dim3(cap_grid((n_nodes * kH + 255) / 256)), dim3(256), 0, s, src, deg, mean,
                     n_nodes, dst);
}

}  // namespace mtmc

#if EK_STAMP
extern "C" int mtmc_dbg_ek_stamps_node(unsigned long long* out) {    // host buffer of 8 * 2 * 8 entries
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mtmc::g_ek), sizeof(mtmc::g_ek));
}
#endif
