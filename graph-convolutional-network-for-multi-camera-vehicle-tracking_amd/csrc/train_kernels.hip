// Backward pass of the message-passing network on gfx950 (training: reference train.py:356,424 runs
// `outputs, _ = mpn_model(data); loss.backward()` through models/mpn.py:250-299).
//
// Everything the backward needs was kept by the training-mode forward in the workspace ("tape"): int32
// row/col, per round z1 / e' / aggregated h, the raw encoder outputs Y_l and every BatchNorm's fp64 sums.
// Pre-activations that were never materialised in the forward (the 32-wide z2, the encoder hiddens) are
// recomputed here from the same inputs with the same Dropout masks (counter-based, common.h).
//
// BatchNorm backward with batch statistics, for y = gamma*zh + beta, zh = (z-mu)*istd over R rows, g = dL/dy:
//     dbeta = sum g,  dgamma = sum g*zh,  dz = gamma*istd*(g - dbeta/R - zh*dgamma/R)
// so every BN layer costs one statistics pass (sum g, sum g*zh in fp64) and one apply pass.
//
// Correctness first: these kernels favour simple, checkable structure (training graphs are ~440 nodes /
// ~180k edges, train.py:277); the forward kernels are the tuned ones.
#include "train_kernels.h"

namespace mtmc {

__device__ __forceinline__ void gacc_add(float* gacc, int slot, float v) {
  unsafeAtomicAdd(gacc + (blockIdx.x % kGradRep) * kGaccN + slot, v);
}

// The small weights an edge kernel needs for EVERY edge, copied to LDS once per workgroup: read through the parameter
// struct they are vector loads from global memory that hipcc cannot hoist or make scalar (the kernels store through
// other pointers), and each one is a memory round trip inside the per-edge code -- 54 `s_waitcnt vmcnt(0)` in
// bwd_edge_upd_kernel<1>, 32 us for 173k edges before this, against 8 us for the forward pass over the same edges.
struct EdgeWeightsLds {
  float w1[8], b1[4], w2[16], b2[4], g1[4], g2[4];     // edge encoder
  float ue_w[32], ue_g[4];                             // edge-update Linear: its 4 x (4|8) edge columns, BN gamma
  float cls_w[4 * MTMC_MAX_CLASSES];
};
__device__ __forceinline__ void edge_weights_to_lds(const EdgeEncParams& enc, const RoundParams* f, EdgeWeightsLds* w) {
  const int t = threadIdx.x;
  if (t < 8) w->w1[t] = t < 4 * enc.fe ? enc.w1[t] : 0.f;
  else if (t < 12) w->b1[t - 8] = enc.b1[t - 8];
  else if (t < 28) w->w2[t - 12] = enc.w2[t - 12];
  else if (t < 32) w->b2[t - 28] = enc.b2[t - 28];
  else if (t < 36) w->g1[t - 32] = enc.g1[t - 32];
  else if (t < 40) w->g2[t - 36] = enc.g2[t - 36];
  else if (f && t >= 64 && t < 96) {
    const int kk = (t - 64) >> 3, j = (t - 64) & 7;
    w->ue_w[t - 64] = j < (f->reattach_edges ? 8 : 4) ? f->ue_w[kk * f->ue_ld + f->ue_eoff + j] : 0.f;
  } else if (f && t >= 96 && t < 100) w->ue_g[t - 96] = f->ue_g[t - 96];
  else if (f && t >= 128 && t < 128 + 4 * MTMC_MAX_CLASSES)
    w->cls_w[t - 128] = (t - 128) < 4 * f->n_classes ? f->cls_w[t - 128] : 0.f;
}
__device__ __forceinline__ EdgeEncParams enc_from_lds(const EdgeEncParams& enc, const EdgeWeightsLds* w) {
  EdgeEncParams e = enc;
  drop_resolve(e.drop);
  e.w1 = w->w1; e.b1 = w->b1; e.w2 = w->w2; e.b2 = w->b2; e.g1 = w->g1; e.g2 = w->g2;
  return e;
}

__device__ __forceinline__ void mean_istd(double sum, double sumsq, double count, float& mu, float& istd) {
  const double inv = 1.0 / count;
  const double mean = sum * inv;
  double var = fma(sumsq, inv, -mean * mean);
  var = (var < 0 ? 0 : var) + MTMC_BN_EPS;
  double r = (double)rsqrtf((float)var);
  r = r * fma(-0.5 * var, r * r, 1.5);
  mu = (float)mean;
  istd = (float)r;
}

// ------------------------------------------------------------------------------------------------
// node-update MLP: statistics (mode 0) and apply (mode 1); lanes = channels as in pass_c_kernel.
// Max aggregation routes a node's gradient to ONE edge per channel, the arg max (torch_scatter.scatter_max
// semantics, reference models/mpn.py:199); mode 2 finds it first: smallest edge index attaining the maximum.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void bwd_node_upd_kernel(BwdRoundParams p) {
  drop_resolve(p.f.drop_n);
  __shared__ float4 tile_e[256];
  __shared__ int tile_row[256];
  __shared__ double st[10 + 64 + 64];     // e' second moments | z2 sum, sumsq | (mode 1) sum g, sum g*zh
  __shared__ double red[8 * 32 * 6];
  // mode 1: dz2 of a half-wave's 32 edges, [edge][channel] with an odd row stride, so that the lanes can swap roles
  // (channel -> edge) and leave A^T.dz2 (4 floats per edge) instead of the 32-wide dz2 itself
  __shared__ float dzt[MODE == 1 ? 8 * 32 * 33 : 1];
  __shared__ float4 a_all[32];
  const int k = threadIdx.x & 31, hw = threadIdx.x >> 5;
  stat_gather2(p.f.stats + kRoundMOff + 4, 10, kMStride, p.f.stats + kRoundZ2Off, 64, kZ2Stride, st);
  if (MODE == 1) stat_gather(p.bst, 64, kBwdStrideD, st + 74);
  __syncthreads();
  const float* aw = p.f.un_w + k * p.f.un_ld + p.f.un_eoff;
  float a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = aw[j];
  if (MODE == 1 && hw == 0) a_all[k] = make_float4(a4[0], a4[1], a4[2], a4[3]);
  const float bk = p.f.un_b[k], gam = p.f.un_g[k], bet = p.f.un_bt[k];
  float mu, istd, sk, tk;
  const double z2sq = st[10 + 32 + k] + quad_form(aw, 4, st);
  mean_istd(st[10 + k], z2sq, p.f.e_total, mu, istd);
  bn_affine(st[10 + k], z2sq, p.f.e_total, gam, bet, sk, tk);     // the forward's affine, bit for bit (pass_c_kernel)
  float a4s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4s[j] = sk * a4[j];
  const float cb = fmaf(sk, bk, tk);
  const float ik = p.f.drop_n.on ? p.f.drop_n.inv_keep : 1.f;
  const float mean_g = MODE == 1 ? (float)(st[74 + k] / p.f.e_total) : 0.f;
  const float mean_gz = MODE == 1 ? (float)(st[74 + 32 + k] / p.f.e_total) : 0.f;
  double acc[6] = {0, 0, 0, 0, 0, 0};      // mode 0: sum g, sum g*zh ; mode 1: sum dz2, sum dz2*e[0..3]

  const int64_t n_tiles = (p.f.n_edges + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t base = tile * 256;
    const int64_t e = base + threadIdx.x;
    if (e < p.f.n_edges) {
      tile_e[threadIdx.x] = reinterpret_cast<const float4*>(p.f.e_out)[e];
      tile_row[threadIdx.x] = p.f.row32[e];
    }
    __syncthreads();
    const int n_here = (int)min((int64_t)32, p.f.n_edges - (base + hw * 32));
    int cur = -1;
    float q = 0.f, dh = 0.f, hval = 0.f, dq_acc = 0.f;
    int winner = -1;
    unsigned long long zd = 0;               // Dropout: the hash of four consecutive edges of this lane's channel (common.h)
    for (int j = 0; j < n_here; ++j) {
      const int r = tile_row[hw * 32 + j];
      const float4 v = tile_e[hw * 32 + j];
      if (p.f.drop_n.on && (j & 3) == 0)
        zd = drop_hash4(p.f.drop_n.seed, p.f.drop_stream + 1, drop_msg_index(base + hw * 32 + j, k, p.f.n_edges) >> 2);
      if (r != cur) {
        if (MODE == 1 && cur >= 0) unsafeAtomicAdd(p.g_Q + (int64_t)cur * kH + k, dq_acc);
        cur = r;
        dq_acc = 0.f;
        q = p.f.Q[(int64_t)r * kH + k];
        dh = p.g_h[(int64_t)r * kH + k];
        if (p.f.agg == 1) { const int d = p.deg[r]; dh = dh / (float)(d > 1 ? d : 1); }
        if (p.f.agg == 2) {
          hval = p.h_agg[(int64_t)r * kH + k];
          if (MODE != 2) winner = p.arg[(int64_t)r * kH + k];
        }
      }
      const int64_t eidx = base + hw * 32 + j;
      const float z2 = fmaf(a4[3], v.w, fmaf(a4[2], v.z, fmaf(a4[1], v.y, fmaf(a4[0], v.x, q + bk))));
      const float zh = (z2 - mu) * istd;
      const float y = fmaf(a4s[3], v.w, fmaf(a4s[2], v.z, fmaf(a4s[1], v.y, fmaf(a4s[0], v.x, fmaf(sk, q, cb)))));
      const bool kept = !p.f.drop_n.on || drop_field(p.f.drop_n, zd, j & 3);
      const bool live = kept && y > 0.f;
      if (MODE == 2) {
        if (live && y * ik == hval) atomicMin(p.arg + (int64_t)r * kH + k, (int)eidx);
        continue;
      }
      float dm = dh;
      if (p.f.agg == 2) dm = (winner == (int)eidx) ? dh : 0.f;
      const float g = live ? dm * ik : 0.f;
      if (MODE == 0) {
        acc[0] += g;
        acc[1] += (double)g * zh;
      } else {
        const float dz2 = gam * istd * (g - mean_g - zh * mean_gz);
        dzt[(hw * 32 + j) * 33 + k] = dz2;
        dq_acc += dz2;
        acc[0] += dz2;
        acc[1] += (double)dz2 * v.x; acc[2] += (double)dz2 * v.y; acc[3] += (double)dz2 * v.z; acc[4] += (double)dz2 * v.w;
      }
    }
    if (MODE == 1 && cur >= 0) unsafeAtomicAdd(p.g_Q + (int64_t)cur * kH + k, dq_acc);
    if (MODE == 1) {
      __builtin_amdgcn_wave_barrier();               // the half-wave's own LDS writes, read back below by other lanes
      if (k < n_here) {                              // lane k now stands for edge k of the half-wave's 32
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
          const float d = dzt[(hw * 32 + k) * 33 + c];
          const float4 a = a_all[c];
          d0 = fmaf(a.x, d, d0); d1 = fmaf(a.y, d, d1); d2 = fmaf(a.z, d, d2); d3 = fmaf(a.w, d, d3);
        }
        reinterpret_cast<float4*>(p.g_de2)[base + hw * 32 + k] = make_float4(d0, d1, d2, d3);
      }
    }
    __syncthreads();
  }
  if (MODE == 2) return;
  constexpr int NV = MODE == 0 ? 2 : 5;
#pragma unroll
  for (int i = 0; i < NV; ++i) red[(i * 8 + hw) * 32 + k] = acc[i];
  __syncthreads();
  if (threadIdx.x < 32) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double s = 0;
      for (int h = 0; h < 8; ++h) s += red[(i * 8 + h) * 32 + k];
      if (MODE == 0) {
        unsafeAtomicAdd(p.bst + (blockIdx.x % kStatRep) * kBwdStrideD + i * 32 + k, s);
      } else if (i == 0) {
        gacc_add(p.gacc, kGaUnB + k, (float)s);
      } else {
        gacc_add(p.gacc, kGaUnW + k * 4 + (i - 1), (float)s);
      }
    }
    if (MODE == 1 && blockIdx.x == 0) {      // dgamma = sum g*zh, dbeta = sum g (the statistics of mode 0)
      p.gr_un_g[k] += (float)st[74 + 32 + k];     // += : the update MLPs are shared by all rounds
      p.gr_un_bt[k] += (float)st[74 + k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge-update MLP + classifier: statistics (mode 0) and apply (mode 1); lanes = edges
// ------------------------------------------------------------------------------------------------
struct EdgeBwdShared { float mu1[4], istd1[4], mean_g[4], mean_gz[4]; double sum_g[4], sum_gz[4]; };

template <int MODE>
__global__ __launch_bounds__(256) void bwd_edge_upd_kernel(BwdRoundParams p) {
  __shared__ EdgeEncAffine af;
  __shared__ EdgeBwdShared sh;
  __shared__ double red[64 * 4];
  __shared__ EdgeWeightsLds wl;
  stat_gather(p.f.stats + kRoundZ1Off, 8, kZ1Stride, red);
  if (MODE == 1) stat_gather(p.bst, 8, kBwdStrideD, red + 8);
  edge_weights_to_lds(p.f.enc, &p.f, &wl);
  const EdgeEncParams enc = enc_from_lds(p.f.enc, &wl);
  const bool need_e0 = MODE == 1 && (p.f.first_round || p.f.reattach_edges);
  if (need_e0) edge_enc_affine_load(p.f.enc, &af); else __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    mean_istd(red[k], red[4 + k], p.f.e_total, sh.mu1[k], sh.istd1[k]);
    if (MODE == 1) {
      sh.sum_g[k] = red[8 + k];
      sh.sum_gz[k] = red[12 + k];
      sh.mean_g[k] = (float)(red[8 + k] / p.f.e_total);
      sh.mean_gz[k] = (float)(red[12 + k] / p.f.e_total);
    }
  }
  __syncthreads();
  const float ik = p.f.drop_e.on ? p.f.drop_e.inv_keep : 1.f;
  const int C = p.f.n_classes;
  constexpr int NV = MODE == 0 ? 8 + 16 + 4 : 4 + 32;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  const int nin = p.f.reattach_edges ? 8 : 4;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e0w = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); e0w < p.f.n_edges; e0w += nthreads) {
    // wave-uniform trip count: mode 1's run reduction is a cross-lane operation every lane has to reach together;
    // lanes past the end compute on the last edge and are masked out of every side effect
    const int64_t e_raw = e0w + (threadIdx.x & 63);
    const bool active = e_raw < p.f.n_edges;
    if (MODE == 0 && !active) continue;
    const int64_t e = active ? e_raw : p.f.n_edges - 1;
    const float4 er4 = reinterpret_cast<const float4*>(p.f.e_out)[e];
    const float4 z4 = reinterpret_cast<const float4*>(p.f.e_buf)[e];
    const float er[4] = {er4.x, er4.y, er4.z, er4.w}, z1[4] = {z4.x, z4.y, z4.z, z4.w};
    float zh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) zh[j] = (z1[j] - sh.mu1[j]) * sh.istd1[j];
    if (MODE == 0) {
      // total gradient wrt e_r: later rounds (g_e) + node-update path (A^T dz2) + classifier (Wc^T dlogits)
      float4 ge4 = reinterpret_cast<const float4*>(p.g_e)[e];
      float de[4] = {ge4.x, ge4.y, ge4.z, ge4.w};
      const float4 d2 = reinterpret_cast<const float4*>(p.g_de2)[e];       // A^T dz2, left by bwd_node_upd_kernel<1>
      de[0] += d2.x; de[1] += d2.y; de[2] += d2.z; de[3] += d2.w;
      if (p.d_logits) {
#pragma unroll
        for (int c = 0; c < MTMC_MAX_CLASSES; ++c) {     // static indices into acc[]: it must stay in registers
          if (c < C) {
            const float dl = p.d_logits[e * C + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              de[j] = fmaf(wl.cls_w[c * 4 + j], dl, de[j]);
              acc[8 + c * 4 + j] += (double)dl * er[j];
            }
            acc[24 + c] += dl;
          }
        }
      }
      float g1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g1[j] = er[j] > 0.f ? de[j] * ik : 0.f;
        acc[j] += g1[j];
        acc[4 + j] += (double)g1[j] * zh[j];
      }
      reinterpret_cast<float4*>(p.g_e)[e] = make_float4(g1[0], g1[1], g1[2], g1[3]);
    } else {
      const float4 g4 = reinterpret_cast<const float4*>(p.g_e)[e];
      const float g1[4] = {g4.x, g4.y, g4.z, g4.w};
      float dz1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dz1[j] = wl.ue_g[j] * sh.istd1[j] * (g1[j] - sh.mean_g[j] - zh[j] * sh.mean_gz[j]);
      const int r = p.f.row32[e], c = p.f.col32[e];
      // dP: ~E/N atomics land on every one of the N*8 addresses and same-address atomics serialise in L2 (measured
      // 53 us of this 62 us kernel); kGradRep replicas by workgroup cut the chains, bwd_node_proj adds them up
      float* gP = p.g_P + (size_t)(blockIdx.x % kGradRep) * p.f.n_nodes * 8;
      // replica layout: dPr [N][4], then dPc [4][N] -- the lanes of a wave hold consecutive (or nearly so) columns, so
      // channel-major puts a wave's 64 column atomics on 2-3 cache lines instead of 16 (17 of this kernel's 34 us were
      // those scattered atomics)
      wave_run_atomic_add<4>(dz1, r, active, gP, 4);      // rows come in long runs: reduce in the wave first
      if (!active) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += dz1[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) unsafeAtomicAdd(gP + (4 + j) * p.f.n_nodes + c, dz1[j]);
      // the edge input of this round: [e0 | e_prev] (reattach) or e_prev, with e_prev = e0 in the first round
      float e0[4] = {0, 0, 0, 0}, ein[8];
      if (p.f.first_round || p.f.reattach_edges) {
        float a0, a1, u[4];
        load_attr(p.f.attr, enc.fe, e, a0, a1);
        edge_enc_hidden(enc, af, e, a0, a1, u);
        edge_enc_out(enc, af, e, u, e0);
      }
      float ep[4];
      if (p.f.first_round) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ep[j] = e0[j];
      } else {
        const float4 v = reinterpret_cast<const float4*>(p.f.e_prev)[e];
        ep[0] = v.x; ep[1] = v.y; ep[2] = v.z; ep[3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { ein[j] = p.f.reattach_edges ? e0[j] : ep[j]; ein[4 + j] = ep[j]; }
      float din[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) {                      // static indices into acc[] / din[] (registers, not scratch)
        if (j < nin) {
          float s = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            s = fmaf(wl.ue_w[kk * 8 + j], dz1[kk], s);
            acc[4 + kk * 8 + j] += (double)dz1[kk] * ein[j];
          }
          din[j] = s;
        }
      }
      // route d e_in: the e_prev part goes to the previous round (or to e0 in the first round), the e0 part to e0
      float d0[4] = {0, 0, 0, 0}, dp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dp[j] = p.f.reattach_edges ? din[4 + j] : din[j];
        if (p.f.reattach_edges) d0[j] = din[j];
      }
      if (p.f.first_round) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d0[j] += dp[j];
      } else {
        reinterpret_cast<float4*>(p.g_e_prev)[e] = make_float4(dp[0], dp[1], dp[2], dp[3]);
      }
      if (p.f.first_round || p.f.reattach_edges) {
        float4 cur0 = reinterpret_cast<float4*>(p.g_e0)[e];
        cur0.x += d0[0]; cur0.y += d0[1]; cur0.z += d0[2]; cur0.w += d0[3];
        reinterpret_cast<float4*>(p.g_e0)[e] = cur0;
      }
    }
  }
  // block reduction of NV doubles
  wave_sums_f64<NV>(acc, red + (threadIdx.x >> 6) * 64);      // (5 instructions per value instead of 18: common.h)
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    const double s = red[i] + red[64 + i] + red[128 + i] + red[192 + i];
    if (MODE == 0) {
      if (i < 8) unsafeAtomicAdd(p.bst + (blockIdx.x % kStatRep) * kBwdStrideD + i, s);
      else if (i < 24) { if ((i - 8) / 4 < C) gacc_add(p.gacc, kGaClsW + (i - 8), (float)s); }
      else if (i - 24 < C) gacc_add(p.gacc, kGaClsB + (i - 24), (float)s);
    } else {
      if (i < 4) gacc_add(p.gacc, kGaUeB + i, (float)s);
      else {
        const int kk = (i - 4) / 8, j = (i - 4) % 8;
        if (j < nin) gacc_add(p.gacc, kGaUeW + kk * 8 + j, (float)s);
      }
    }
  }
  if (MODE == 1 && blockIdx.x == 0 && threadIdx.x < 4) {
    p.gr_ue_g[threadIdx.x] += (float)sh.sum_gz[threadIdx.x];   // += : shared by all rounds
    p.gr_ue_bt[threadIdx.x] += (float)sh.sum_g[threadIdx.x];
  }
}

// ------------------------------------------------------------------------------------------------
// node projection backward: d[h0|h] and the node-column blocks of the two update weights
// ------------------------------------------------------------------------------------------------
constexpr int kBwdProjNodes = 16;   // nodes per workgroup: 430 nodes must still give a few dozen workgroups (32: 14 us per
                                    // launch on 14 CUs); the weight-gradient atomics meet N/16 deep on a word
__global__ __launch_bounds__(256) void bwd_node_proj_kernel(BwdProjParams p) {
  constexpr int NB = kBwdProjNodes;
  __shared__ float hs[NB * 65];
  __shared__ float gs[NB * 41];
  const int hn = p.hn, ldh = hn + 1;
  const int64_t n_groups = (p.n_nodes + NB - 1) / NB;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t node0 = g * NB;
    __syncthreads();
    for (int i = threadIdx.x; i < NB * kH; i += blockDim.x) {
      const int n = i >> 5, kk = i & 31;
      const int64_t node = node0 + n;
      float v = 0.f, v0 = 0.f;
      if (node < p.n_nodes) {
        v = p.h_src[node * kH + kk];
        if (p.deg) { const int d = p.deg[node]; v = v / (float)(d > 1 ? d : 1); }
        if (hn == 2 * kH) v0 = p.h0[node * kH + kk];
      }
      hs[n * ldh + (hn - kH) + kk] = v;
      if (hn == 2 * kH) hs[n * ldh + kk] = v0;
    }
    for (int i = threadIdx.x; i < NB * 40; i += blockDim.x) {
      const int n = i / 40, j = i % 40;
      const int64_t node = node0 + n;
      float v = 0.f;
      if (node < p.n_nodes) {
        if (j < 8) {
#pragma unroll
          for (int rep = 0; rep < kGradRep; ++rep)
            v += p.g_P[(size_t)rep * p.n_nodes * 8 + (j < 4 ? node * 4 + j : (size_t)j * p.n_nodes + node)];
        } else {
          v = p.g_Q[node * kH + j - 8];
        }
      }
      gs[n * 41 + j] = v;
    }
    __syncthreads();
    // d[h0|h][node][c] = sum_j g[node][j] * W_j[c]
    for (int i = threadIdx.x; i < NB * hn; i += blockDim.x) {
      const int n = i / hn, c = i % hn;
      const int64_t node = node0 + n;
      if (node >= p.n_nodes) continue;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s = fmaf(gs[n * 41 + j], p.ue_w[j * p.ue_ld + c], s);
        s = fmaf(gs[n * 41 + 4 + j], p.ue_w[j * p.ue_ld + hn + c], s);
      }
      for (int kk = 0; kk < kH; ++kk) s = fmaf(gs[n * 41 + 8 + kk], p.un_w[kk * p.un_ld + c], s);
      const bool is_h0 = (hn == 2 * kH && c < kH) || p.src_is_h0;
      const int cc = c & 31;
      if (is_h0) {
        // [h0|h0] in the first round of a reattach model: both halves land on the same element
        unsafeAtomicAdd(p.g_h0 + node * kH + cc, s);
      } else {
        p.g_h_prev[node * kH + cc] = s;
      }
    }
    // weight gradients: 40 x hn outputs, each summed over this block's NB nodes
    for (int i = threadIdx.x; i < 40 * hn; i += blockDim.x) {
      const int j = i / hn, c = i % hn;
      float s = 0.f;
      for (int n = 0; n < NB; ++n) s = fmaf(gs[n * 41 + j], hs[n * ldh + c], s);
      float* dst = j < 4 ? p.gr_ue_w + j * p.ue_ld + c
                         : (j < 8 ? p.gr_ue_w + (j - 4) * p.ue_ld + hn + c : p.gr_un_w + (j - 8) * p.un_ld + c);
      unsafeAtomicAdd(dst, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge encoder backward (three passes over edge_attr; everything is recomputed from the 8-byte attributes)
// ------------------------------------------------------------------------------------------------
struct EncBwdShared { float mua[4], ia[4], mub[4], ib[4], mgb[4], mgzb[4], mga[4], mgza[4]; double sums[16]; };

// mu/istd of an affine layer's pre-activation from the packed moments of its input (cf. moments_affine)
__device__ __forceinline__ void moments_mean_istd(const float* w, int in_dim, float bias, const double* m1,
                                                  const double* m2, double count, float& mu, float& istd) {
  double wm1 = 0;
  for (int i = 0; i < in_dim; ++i) wm1 += (double)w[i] * m1[i];
  const double q = quad_form(w, in_dim, m2), b = bias;
  mean_istd(wm1 + b * count, q + 2.0 * b * wm1 + b * b * count, count, mu, istd);
}

template <int PASS>   // 0: stats of layer 2; 1: apply layer 2 + stats of layer 1; 2: apply layer 1
__global__ __launch_bounds__(256) void bwd_edge_enc_kernel(BwdEncParams p) {
  drop_resolve(p.enc.drop);
  __shared__ EdgeEncAffine af;
  __shared__ EncBwdShared sh;
  __shared__ double sc[kStatAttr + kStatEnc2 + 16];
  __shared__ double red[64 * 4];
  __shared__ EdgeWeightsLds wl;
  edge_weights_to_lds(p.enc, nullptr, &wl);                       // (the first barrier below publishes it)
  edge_enc_affine_to_smem(p.enc, p.e_total, 2, &af, sc);          // also leaves the summed moments in sc
  if (PASS >= 1) stat_gather(p.bst, 8, kBwdStrideD, sc + kStatAttr + kStatEnc2);
  if (PASS >= 2) stat_gather(p.bst + 8, 8, kBwdStrideD, sc + kStatAttr + kStatEnc2 + 8);
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    moments_mean_istd(p.enc.w1 + k * p.enc.fe, p.enc.fe, p.enc.b1[k], sc, sc + 2, p.e_total, sh.mua[k], sh.ia[k]);
    moments_mean_istd(p.enc.w2 + k * 4, 4, p.enc.b2[k], sc + kStatAttr, sc + kStatAttr + 4, p.e_total, sh.mub[k], sh.ib[k]);
    const double* b = sc + kStatAttr + kStatEnc2;
    if (PASS >= 1) {
      sh.mgb[k] = (float)(b[k] / p.e_total); sh.mgzb[k] = (float)(b[4 + k] / p.e_total);
      for (int i = 0; i < 4; ++i) sh.sums[i * 4 + k] = b[i * 4 + k];
    }
    if (PASS >= 2) { sh.mga[k] = (float)(b[8 + k] / p.e_total); sh.mgza[k] = (float)(b[12 + k] / p.e_total); }
  }
  __syncthreads();
  const float ik = p.enc.drop.on ? p.enc.drop.inv_keep : 1.f;
  constexpr int NV = PASS == 0 ? 8 : (PASS == 1 ? 8 + 16 + 4 : 8 + 4);
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  const int fe = p.enc.fe;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < p.n_edges; e += nthreads) {
    float a0, a1, za[4], u[4], zb[4], e0[4], zha[4], zhb[4];
    load_attr(p.attr, fe, e, a0, a1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float z = wl.b1[k] + wl.w1[k * fe] * a0;
      if (fe > 1) z = fmaf(wl.w1[k * fe + 1], a1, z);
      za[k] = z;
      zha[k] = (z - sh.mua[k]) * sh.ia[k];
      u[k] = fmaxf(fmaf(z, af.s1[k], af.t1[k]), 0.f);
    }
    drop_apply4(p.enc.drop, kDropEncEdge1, (unsigned long long)e * 4, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float z = wl.b2[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) z = fmaf(wl.w2[k * 4 + j], u[j], z);
      zb[k] = z;
      zhb[k] = (z - sh.mub[k]) * sh.ib[k];
      e0[k] = fmaxf(fmaf(z, af.s2[k], af.t2[k]), 0.f);
    }
    drop_apply4(p.enc.drop, kDropEncEdge2, (unsigned long long)e * 4, e0);
    const float4 d4 = reinterpret_cast<const float4*>(p.g_e0)[e];
    const float de0[4] = {d4.x, d4.y, d4.z, d4.w};
    float gb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gb[k] = e0[k] > 0.f ? de0[k] * ik : 0.f;
    if (PASS == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc[k] += gb[k]; acc[4 + k] += (double)gb[k] * zhb[k]; }
      continue;
    }
    float dzb[4], du[4] = {0, 0, 0, 0}, ga[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) dzb[k] = wl.g2[k] * sh.ib[k] * (gb[k] - sh.mgb[k] - zhb[k] * sh.mgzb[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) du[j] = fmaf(wl.w2[k * 4 + j], dzb[k], du[j]);
#pragma unroll
    for (int k = 0; k < 4; ++k) ga[k] = u[k] > 0.f ? du[k] * ik : 0.f;
    if (PASS == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] += ga[k];
        acc[4 + k] += (double)ga[k] * zha[k];
        acc[24 + k] += dzb[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[8 + k * 4 + j] += (double)dzb[k] * u[j];
      }
      continue;
    }
    float dza[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dza[k] = wl.g1[k] * sh.ia[k] * (ga[k] - sh.mga[k] - zha[k] * sh.mgza[k]);
      acc[8 + k] += dza[k];
      acc[k * 2] += (double)dza[k] * a0;
      acc[k * 2 + 1] += (double)dza[k] * a1;
    }
    if (p.d_attr) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { s0 = fmaf(wl.w1[k * fe], dza[k], s0); if (fe > 1) s1 = fmaf(wl.w1[k * fe + 1], dza[k], s1); }
      p.d_attr[e * fe] = s0;
      if (fe > 1) p.d_attr[e * fe + 1] = s1;
    }
    (void)za; (void)zb;
  }
  wave_sums_f64<NV>(acc, red + (threadIdx.x >> 6) * 64);      // (5 instructions per value instead of 18: common.h)
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    const double s = red[i] + red[64 + i] + red[128 + i] + red[192 + i];
    if (PASS == 0) unsafeAtomicAdd(p.bst + (blockIdx.x % kStatRep) * kBwdStrideD + i, s);
    else if (PASS == 1) {
      if (i < 8) unsafeAtomicAdd(p.bst + (blockIdx.x % kStatRep) * kBwdStrideD + 8 + i, s);
      else if (i < 24) gacc_add(p.gacc, kGaW2 + (i - 8), (float)s);
      else gacc_add(p.gacc, kGaB2 + (i - 24), (float)s);
    } else {
      if (i < 8) { if ((i & 1) < fe) gacc_add(p.gacc, kGaW1 + i, (float)s); }
      else gacc_add(p.gacc, kGaB1 + (i - 8), (float)s);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    const int k = threadIdx.x;
    if (PASS == 1) { p.gr_g2[k] = (float)sh.sums[4 + k]; p.gr_bt2[k] = (float)sh.sums[k]; }
    if (PASS == 2) { p.gr_g1[k] = (float)sh.sums[12 + k]; p.gr_bt1[k] = (float)sh.sums[8 + k]; }
  }
}

// ------------------------------------------------------------------------------------------------
// dense helpers of the node-encoder backward
// ------------------------------------------------------------------------------------------------
// column statistics of g = (a > 0 ? dA/keep : 0) against zh = (Y-mu)*istd: stats[0..d) = sum g, [d..2d) = sum g*zh;
// MODE 1: dY = gamma*istd*(g - mean_g - zh*mean_gz) written over dA, and column sums of dY -> d_bias
constexpr int kBnBwdRows = 16;   // rows per workgroup: a few hundred node rows must still spread over the chip (64 rows
                                 // per workgroup: 12.4 / 8.1 us per launch at 430 x 1024)
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdParams p) {
  if (MODE == 1 && blockIdx.z == 1) {               // passenger: relu(bn(Y_{l-1})) and its transpose (rows_body.h)
    RowsTJob j = p.rc;
    drop_resolve(j.drop);
    if ((int)blockIdx.x * 64 < j.dim) bn_relu_rows_t_body(j, blockIdx.x, blockIdx.y, gridDim.y);
    return;
  }
  if ((int)blockIdx.x * 64 >= p.dim) return;        // (the grid is as wide as the wider of the two jobs)
  drop_resolve(p.drop);
  __shared__ double red[2 * 4 * 64];
  __shared__ float tile[MODE == 1 ? kBnBwdRows : 1][65];          // MODE 1: dY^T leaves as 16-byte stores, one per thread
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * kBnBwdRows;
  double s0 = 0, s1 = 0;
  float dymax = 0.f;
  if (col < p.dim) {
    float mu, istd;
    mean_istd(p.stats_fwd[col], p.stats_fwd[p.dim + col], p.count, mu, istd);
    const float gam = p.gamma[col], bet = p.beta[col];
    const float ik = p.drop.on ? p.drop.inv_keep : 1.f;
    const float mg = MODE == 1 ? (float)(p.stats_bwd[col] / p.count) : 0.f;
    const float mgz = MODE == 1 ? (float)(p.stats_bwd[p.dim + col] / p.count) : 0.f;
#pragma unroll
    for (int i = 0; i < kBnBwdRows / 4; ++i) {
      const int64_t row = r0 + rg + 4 * i;
      if (MODE == 1) tile[rg + 4 * i][cl] = 0.f;                  // (rows past the end: the transposed copy's zero padding)
      if (row >= p.rows) continue;
      const float zh = (p.Y[row * p.dim + col] - mu) * istd;
      const float y = fmaf(gam, zh, bet);
      const bool live = y > 0.f && drop_keep(p.drop, p.drop_stream, (unsigned long long)row * p.dim + col);
      const float g = live ? p.dA[row * p.dim + col] * ik : 0.f;
      if (MODE == 0) {
        s0 += g;
        s1 += (double)g * zh;
      } else {
        const float dy = gam * istd * (g - mg - zh * mgz);
        p.dA[row * p.dim + col] = dy;
        tile[rg + 4 * i][cl] = dy;
        dymax = fmaxf(dymax, fabsf(dy));
        s0 += dy;
      }
    }
  }
  if (MODE == 1 && p.dT && blockIdx.y == gridDim.y - 1 && col < p.dim)      // the transposed copy's padding rows behind the last tile
    for (int64_t row = r0 + kBnBwdRows + rg; row < p.ldt; row += 4) p.dT[(int64_t)col * p.ldt + row] = 0.f;
  red[rg * 64 + cl] = s0;
  red[256 + rg * 64 + cl] = s1;
  __syncthreads();
  if (MODE == 1 && p.dT) {
    static_assert(kBnBwdRows == 16, "a thread stores a quarter of a column's 16 rows");
    const int c = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (blockIdx.x * 64 + c < p.dim && r0 + 4 * q < p.ldt)        // (ldt: the row count padded to 32)
      *reinterpret_cast<float4*>(p.dT + (int64_t)(blockIdx.x * 64 + c) * p.ldt + r0 + 4 * q) =
          make_float4(tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]);
  }
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c < p.dim) {
      const double* q = red + which * 256 + (threadIdx.x & 63);
      const double s = q[0] + q[64] + q[128] + q[192];
      if (MODE == 0) unsafeAtomicAdd(p.stats_bwd + which * p.dim + c, s);
      else if (which == 0) unsafeAtomicAdd(p.gr_bias + c, (float)s);
    }
  }
  if (MODE == 1 && p.amax_out) {                   // operand scale of the weight- / input-gradient GEMMs
    __shared__ float wmax[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dymax = fmaxf(dymax, __shfl_xor(dymax, off, 64));
    if (cl == 0) wmax[rg] = dymax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_publish(p.amax_out + (blockIdx.x + blockIdx.y) % kAmaxRep, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
  if (MODE == 1 && blockIdx.y == 0 && threadIdx.x < 64 && col < p.dim) {
    p.gr_gamma[col] = (float)p.stats_bwd[p.dim + col];
    p.gr_beta[col] = (float)p.stats_bwd[col];
  }
}

// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < rows_pad   (dst leading dimension rows_pad)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* src, int64_t rows, int cols, int64_t ld_src,
                                                            float* dst, int64_t rows_pad) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int64_t r = r0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int64_t r = r0 + tx;
    if (c < cols && r < rows_pad) dst[(int64_t)c * rows_pad + r] = tile[tx][i];
  }
}

// The same for several matrices in ONE launch (the node encoder's backward needs x^T and every W_l^T, all of them known before
// its first kernel: four 5 us launches on the chain become one).  1-D grid; a block finds its matrix by the block prefix.
__device__ __forceinline__ void transpose_job_body(const TransposeJobs& p, unsigned block) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < p.n && block >= p.job[j + 1].first_block) ++j;
  const TransposeJob& q = p.job[j];
  const unsigned b = block - q.first_block;
  const int64_t r0 = (int64_t)(b % q.blocks_r) * 32;
  const int c0 = (int)(b / q.blocks_r) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int64_t r = r0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (r < q.rows && c < q.cols) ? q.src[r * q.ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int64_t r = r0 + tx;
    if (c < q.cols && r < q.rows_pad) q.dst[(int64_t)c * q.rows_pad + r] = tile[tx][i];
  }
}
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeJobs p) { transpose_job_body(p, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// the replicated small-gradient sums -> the caller's gradient tensors (one workgroup, the last launch of the backward;
// plain += : nothing else touches these words any more, the node-column blocks of the update weights are other words)
__global__ __launch_bounds__(256) void grad_fold_kernel(GradFoldParams p) {
  const int i = threadIdx.x;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < kGradRep; ++r) s += p.gacc[r * kGaccN + i];
  float* dst = nullptr;
  if (i < kGaUnW) dst = p.gr_un_b + i;
  else if (i < kGaClsW) dst = p.gr_un_w + ((i - kGaUnW) >> 2) * p.un_ld + p.un_eoff + ((i - kGaUnW) & 3);
  else if (i < kGaClsB) { if ((i - kGaClsW) / 4 < p.n_classes) dst = p.gr_cls_w + (i - kGaClsW); }
  else if (i < kGaUeB) { if (i - kGaClsB < p.n_classes) dst = p.gr_cls_b + (i - kGaClsB); }
  else if (i < kGaUeW) dst = p.gr_ue_b + (i - kGaUeB);
  else if (i < kGaW2) { const int kk = (i - kGaUeW) >> 3, j = (i - kGaUeW) & 7; if (j < p.nin) dst = p.gr_ue_w + kk * p.ue_ld + p.ue_eoff + j; }
  else if (i < kGaB2) dst = p.gr_w2 + (i - kGaW2);
  else if (i < kGaW1) dst = p.gr_b2 + (i - kGaB2);
  else if (i < kGaB1) { const int q = i - kGaW1; if ((q & 1) < p.fe) dst = p.gr_w1 + (q >> 1) * p.fe + (q & 1); }
  else if (i < kGaB1 + 4) dst = p.gr_b1 + (i - kGaB1);
  if (dst) *dst += s;
}
void launch_grad_fold(const GradFoldParams& p, hipStream_t s) {
  hipLaunchKernelGGL(grad_fold_kernel, dim3(1), dim3(kGaccN), 0, s, p);
}

static inline int cap(int64_t blocks, int64_t hi = 1024) { return (int)(blocks < 1 ? 1 : (blocks > hi ? hi : blocks)); }

void launch_bwd_node_upd(const BwdRoundParams& p, int mode, hipStream_t s) {
  const int grid = cap((p.f.n_edges + 255) / 256);
  if (mode == 0) hipLaunchKernelGGL(bwd_node_upd_kernel<0>, dim3(grid), dim3(256), 0, s, p);
  else if (mode == 1) hipLaunchKernelGGL(bwd_node_upd_kernel<1>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bwd_node_upd_kernel<2>, dim3(grid), dim3(256), 0, s, p);
}
void launch_bwd_edge_upd(const BwdRoundParams& p, int mode, hipStream_t s) {
  const int grid = cap((p.f.n_edges + 255) / 256);
  if (mode == 0) hipLaunchKernelGGL(bwd_edge_upd_kernel<0>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bwd_edge_upd_kernel<1>, dim3(grid), dim3(256), 0, s, p);
}
void launch_bwd_node_proj(const BwdProjParams& p, hipStream_t s) {
  hipLaunchKernelGGL(bwd_node_proj_kernel, dim3(cap((p.n_nodes + kBwdProjNodes - 1) / kBwdProjNodes)), dim3(256), 0, s, p);
}
// L == 0 (reference models/mpn.py:295-297): the classifier sits directly on the encoded edges.
// d e0 = Wc^T d logits; dWc = sum_e d logits (x) e0; dbc = sum_e d logits.
__global__ __launch_bounds__(256) void bwd_classify_e0_kernel(EdgeEncParams enc, const float* attr, int64_t n_edges,
                                                              double e_total, const float* cls_w, int n_classes,
                                                              const float* d_logits, float* g_e0, float* gr_cls_w,
                                                              float* gr_cls_b) {
  drop_resolve(enc.drop);
  __shared__ EdgeEncAffine af;
  __shared__ double scratch[kStatAttr + kStatEnc2];
  __shared__ double red[64 * 4];
  edge_enc_affine_to_smem(enc, e_total, 2, &af, scratch);
  constexpr int NV = 4 * MTMC_MAX_CLASSES + MTMC_MAX_CLASSES;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += nthreads) {
    float a0, a1, u[4], e0[4], de[4] = {0, 0, 0, 0};
    load_attr(attr, enc.fe, e, a0, a1);
    edge_enc_hidden(enc, af, e, a0, a1, u);
    edge_enc_out(enc, af, e, u, e0);
#pragma unroll
    for (int c = 0; c < MTMC_MAX_CLASSES; ++c) {
      if (c < n_classes) {
        const float dl = d_logits[e * n_classes + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          de[j] = fmaf(cls_w[c * 4 + j], dl, de[j]);
          acc[c * 4 + j] += (double)dl * e0[j];
        }
        acc[4 * MTMC_MAX_CLASSES + c] += dl;
      }
    }
    reinterpret_cast<float4*>(g_e0)[e] = make_float4(de[0], de[1], de[2], de[3]);
  }
  wave_sums_f64<NV>(acc, red + (threadIdx.x >> 6) * 64);      // (5 instructions per value instead of 18: common.h)
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    const double s = red[i] + red[64 + i] + red[128 + i] + red[192 + i];
    if (i < 4 * MTMC_MAX_CLASSES) { if (i / 4 < n_classes) unsafeAtomicAdd(gr_cls_w + i, (float)s); }
    else if (i - 4 * MTMC_MAX_CLASSES < n_classes) unsafeAtomicAdd(gr_cls_b + (i - 4 * MTMC_MAX_CLASSES), (float)s);
  }
}

void launch_bwd_classify_e0(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total, const float* cls_w,
                            int n_classes, const float* d_logits, float* g_e0, float* gr_cls_w, float* gr_cls_b,
                            hipStream_t s) {
  hipLaunchKernelGGL(bwd_classify_e0_kernel, dim3(cap((n_edges + 255) / 256)), dim3(256), 0, s, enc, attr, n_edges, e_total,
                     cls_w, n_classes, d_logits, g_e0, gr_cls_w, gr_cls_b);
}
void launch_bwd_edge_enc(const BwdEncParams& p, int pass, hipStream_t s) {
  const int grid = cap((p.n_edges + 255) / 256);
  if (pass == 0) hipLaunchKernelGGL(bwd_edge_enc_kernel<0>, dim3(grid), dim3(256), 0, s, p);
  else if (pass == 1) hipLaunchKernelGGL(bwd_edge_enc_kernel<1>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bwd_edge_enc_kernel<2>, dim3(grid), dim3(256), 0, s, p);
}
void launch_bn_bwd(const BnBwdParams& p, int mode, hipStream_t s) {
  dim3 grid((p.dim + 63) / 64, (unsigned)((p.rows + kBnBwdRows - 1) / kBnBwdRows));
  if (mode == 0) { hipLaunchKernelGGL(bn_bwd_kernel<0>, grid, dim3(256), 0, s, p); return; }
  if (p.rc_on) {                                     // + the recomputation job: z = 1, the wider of the two in x
    static_assert(kBnBwdRows == 16, "both jobs cut the rows in 16s");
    const unsigned xb = (unsigned)(p.rc.dim + 63) / 64;
    grid.x = grid.x > xb ? grid.x : xb;
    grid.z = 2;
  }
  hipLaunchKernelGGL(bn_bwd_kernel<1>, grid, dim3(256), 0, s, p);
}
bool bn_bwd_carries_rows_job(int64_t rows, int64_t ldt) { return ldt % 4 == 0 && rows > 0 && (rows + 15) / 16 <= 65535; }
void launch_transpose_pad(const float* src, int64_t rows, int cols, int64_t ld_src, float* dst, int64_t rows_pad,
                          hipStream_t s) {
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((rows_pad + 31) / 32), (cols + 31) / 32), dim3(256), 0, s,
                     src, rows, cols, ld_src, dst, rows_pad);
}

__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges z) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < z.n; ++j)
    for (size_t i = t0; i < z.r[j].n16; i += nthreads) z.r[j].p[i] = make_uint4(0u, 0u, 0u, 0u);
}
// The backward's first launch: what it accumulates into is cleared by the first n_zero workgroups, x^T and every W_l^T (which
// depend on nothing the backward computes) are made by the rest.
__global__ __launch_bounds__(256) void bwd_begin_kernel(ZeroRanges z, TransposeJobs t, unsigned n_zero) {
  if (blockIdx.x < n_zero) {
    const size_t nthreads = (size_t)n_zero * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < z.n; ++j)
      for (size_t i = t0; i < z.r[j].n16; i += nthreads) z.r[j].p[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  transpose_job_body(t, blockIdx.x - n_zero);
}
void launch_bwd_begin(const ZeroRanges& z, const TransposeJobs& t, hipStream_t s) {
  size_t total = 0;
  for (int j = 0; j < z.n; ++j) total += z.r[j].n16;
  const size_t blocks = (total + 1023) / 1024;
  const unsigned nz = (unsigned)(blocks > 2048 ? 2048 : blocks);
  if (nz + t.n_blocks == 0) return;
  hipLaunchKernelGGL(bwd_begin_kernel, dim3(nz + t.n_blocks), dim3(256), 0, s, z, t, nz);
}
void launch_zero_ranges(const ZeroRanges& z, hipStream_t s) {
  size_t total = 0;
  for (int j = 0; j < z.n; ++j) total += z.r[j].n16;
  if (total == 0) return;
  const size_t blocks = (total + 1023) / 1024;                     // four 16-byte stores per thread
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, z);
}

void transpose_jobs_add(TransposeJobs& p, const float* src, int64_t rows, int cols, int64_t ld_src, float* dst, int64_t rows_pad) {
  TransposeJob& q = p.job[p.n];
  q.src = src; q.rows = rows; q.cols = cols; q.ld_src = ld_src; q.dst = dst; q.rows_pad = rows_pad;
  q.blocks_r = (unsigned)((rows_pad + 31) / 32);
  q.first_block = p.n_blocks;
  p.n_blocks += q.blocks_r * (unsigned)((cols + 31) / 32);
  ++p.n;
}
void launch_transpose_multi(const TransposeJobs& p, hipStream_t s) {
  if (p.n_blocks == 0) return;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(p.n_blocks), dim3(256), 0, s, p);
}

}  // namespace mtmc
