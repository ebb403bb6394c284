// Per-node kernels of the message-passing rounds (tiny next to the edge passes: O(N*32*40) flops).
//
//   node_proj_kernel  Pr|Pc = [h0|h] . We[:, node cols]^T  (4+4 per node),  Q = [h0|h] . Wn[:, node cols]^T (32)
//                     -- the algebraic form of the reference's x[row], x[col] gathers + cat + Linear
//                        (reference models/mpn.py:48,68 and :97-98): gather 16 B per edge end instead of 128 B
//   node_stat_kernel  sum / sum of squares over all E edges of z2 = Q[row] + A.e' + b, from per-node segment
//                     sums of e' and the degree (no pass over the edges)
//   bn_relu_rows      h0 = relu(bn(Y_last))
//   h_final           latent_node_feats output (mean aggregation divides by max(deg,1))
#include "kernels.h"

namespace mtmc {

constexpr int kProjNodes = 16;   // nodes per block iteration (16 lanes per node: 8 output groups x 2 halves of the k range)
constexpr int kProjOut = 40;     // 4 (Pr) + 4 (Pc) + 32 (Q)

__global__ __launch_bounds__(256) void node_proj_kernel(NodeProjParams p) {
  __shared__ float wt[64 * kProjOut];              // [k][j], k < hn
  __shared__ float hs[kProjNodes * 66];            // [node][k], row stride hn+2 (see the k loop)
  __shared__ float ys[kH], yt[kH];                 // BatchNorm affine of the encoder's last layer (fused round 0)
  __shared__ EdgeEncAffine enc_af;
  __shared__ double scratch[kStatAttr + kStatEnc2];
  const int hn = p.hn, ldh = hn + 2;
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
  const int64_t n_groups = (p.node_end - p.node_begin + kProjNodes - 1) / kProjNodes;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t node0 = p.node_begin + g * kProjNodes;
    __syncthreads();                               // wt/ys ready / previous hs consumed
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
    }
  }
  if (p.finalize_enc && blockIdx.x == 0) {         // once per forward: the edge encoder's two BatchNorm affines
    edge_enc_affine_to_smem(p.enc, p.e_total, 2, &enc_af, scratch);
    if (threadIdx.x < 16) p.enc.aff[threadIdx.x] = reinterpret_cast<const float*>(&enc_af)[threadIdx.x];
  }
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
  hipLaunchKernelGGL(node_proj_kernel, dim3(cap_grid((p.node_end - p.node_begin + kProjNodes - 1) / kProjNodes)), dim3(256), 0, s, p);
}
void launch_node_stat(const NodeStatParams& p, hipStream_t s) {
  hipLaunchKernelGGL(node_stat_kernel, dim3(cap_grid((p.node_end - p.node_begin + 7) / 8)), dim3(256), 0, s, p);
}
void launch_bn_relu_rows(const float* Y, int64_t ldy, int64_t rows, int dim, const double* stats, const float* gamma,
                         const float* beta, double count, float* dst, Drop drop, unsigned drop_stream, int64_t row0,
                         hipStream_t s, unsigned* amax_out, float* dstT, int64_t ldt) {
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
