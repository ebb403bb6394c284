// Shared device helpers for the gfx950 message-passing kernels.
// gfx950 only: 64-wide wavefronts are hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MTMC_BN_EPS 1e-5

namespace mtmc {

constexpr int kWave = 64;
constexpr int kH = 32;    // node state width
constexpr int kHe = 4;    // edge state width

// ---- statistics layout (doubles) -------------------------------------------------------------
// Second-moment matrices are symmetric and stored packed (upper triangle, row-major):
//   tri(n,i,j), i<=j  ->  i*n - i*(i-1)/2 + (j-i)
// attr moments : m1[2] | m2 packed[3]                                   (6 doubles reserved)
// enc2 moments : m1[4] | m2 packed[10]                                  (16 reserved)
// round stats  : z1 sum[4] | z1 sumsq[4] | e' m1[4] | e' m2 packed[10] | z2 sum[32] | z2 sumsq[32]
constexpr int kStatAttr = 6;
constexpr int kStatEnc2 = 16;
// a round's statistics are three separately replicated blocks (each contiguous, so a multi-GPU host can
// all-reduce them one at a time): z1 sum[4]|sumsq[4]  ;  e' m1[4]|m2 packed[10]  ;  z2 sum[32]|sumsq[32]
// Every such block is kept in kStatRep replicas (each padded to a 128-byte multiple): a workgroup adds its
// partial sums to replica blockIdx % kStatRep, the consumer adds the replicas up.  Hundreds of workgroups
// adding to ONE cache line serialise at the memory side (float atomics run an order of magnitude slower
// on a single row, MI355X_MICROARCH 'Global float atomics'); 16 lines remove that.
constexpr int kStatRep = 16;
__host__ __device__ __forceinline__ constexpr int stat_stride(int n) { return (n + 15) / 16 * 16; }
constexpr int kAttrStride = stat_stride(kStatAttr), kEnc2Stride = stat_stride(kStatEnc2);   // doubles per replica
constexpr int kZ1Stride = 16, kMStride = 16, kZ2Stride = 64;
constexpr int kRoundZ1Off = 0, kRoundMOff = kStatRep * kZ1Stride, kRoundZ2Off = kRoundMOff + kStatRep * kMStride;
constexpr int kRoundBlock = kRoundZ2Off + kStatRep * kZ2Stride;   // doubles per round

__host__ __device__ __forceinline__ constexpr int tri(int n, int i, int j) { return i * n - i * (i - 1) / 2 + (j - i); }

// Wave-wide sum on the DPP data path (no LDS round trips): inclusive scan inside each row of 16 lanes
// (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals up; lane 63 ends with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add_f64<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add_f64<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add_f64<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add_f64<0x118, 0xf>(v);   // row_shr:8
  v = dpp_add_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;                          // lane 63 holds the total
}
constexpr int kWaveSumLane = 63;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {   // total in lane 63
  v = dpp_add_f32<0x111, 0xf>(v);
  v = dpp_add_f32<0x112, 0xf>(v);
  v = dpp_add_f32<0x114, 0xf>(v);
  v = dpp_add_f32<0x118, 0xf>(v);
  v = dpp_add_f32<0x142, 0xa>(v);
  v = dpp_add_f32<0x143, 0xc>(v);
  return v;
}

// Wave totals of FOUR per-lane values in ten instructions (four separate DPP sums take 24): two v_permlane32_swap + adds
// fold the wave's halves so that lanes 0-31 hold v0 / v2 and lanes 32-63 v1 / v3 partials, one v_permlane16_swap + add
// folds the 16-lane rows, four row_shr steps finish inside a row.  The total of value {0, 2, 1, 3}[r] ends up in the LAST
// lane of row r (lanes 15, 31, 47, 63); wave_sum4_slot(lane) names the value a lane holds.
__device__ __forceinline__ float wave_sum4_f32(float v0, float v1, float v2, float v3) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
  const float s01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // lanes 0-31: v0[l] + v0[l+32]; lanes 32-63: the same of v1
  const float s23 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
  float u = __uint_as_float(c[0]) + __uint_as_float(c[1]);           // rows 0..3: v0, v2, v1, v3 (16 partial sums each)
  u = dpp_add_f32<0x111, 0xf>(u);
  u = dpp_add_f32<0x112, 0xf>(u);
  u = dpp_add_f32<0x114, 0xf>(u);
  u = dpp_add_f32<0x118, 0xf>(u);
  return u;
}
__device__ __forceinline__ int wave_sum4_slot(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }   // rows 0..3 -> 0, 2, 1, 3

// dst[key*stride + j] += v[j] for every active lane, with lanes that carry the same key next to each other reduced
// in the wave first: one DPP sum + NV atomics when the whole wave shares a key (the common case on row-sorted edge
// lists, where per-lane atomics would all hit one address), otherwise a segmented scan and one atomic per run.
// All 64 lanes must call (inactive lanes: active = false; their v is ignored).
template <int NV>
__device__ __forceinline__ void wave_run_atomic_add(const float (&vin)[NV], int key, bool active, float* dst,
                                                    int64_t stride) {
  const int lane = threadIdx.x & 63;
  if (!active) key = -1;
  float v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = active ? vin[j] : 0.f;
  const int k0 = __builtin_amdgcn_readfirstlane(key);
  if (__all(key == k0)) {
    if (k0 < 0) return;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = wave_sum_f32(v[j]);
    if (lane == kWaveSumLane) {
#pragma unroll
      for (int j = 0; j < NV; ++j) unsafeAtomicAdd(dst + (int64_t)k0 * stride + j, v[j]);
    }
    return;
  }
  const int prev = __shfl_up(key, 1, 64);
  int flag = (lane == 0 || prev != key) ? 1 : 0;
  const int next_head = __shfl_down(flag, 1, 64);
  const bool tail = active && (lane == 63 || next_head != 0);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int f_up = __shfl_up(flag, off, 64);
    float u[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) u[j] = __shfl_up(v[j], off, 64);
    if (lane >= off && !flag) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] += u[j];
      flag = f_up;
    }
  }
  if (tail) {
#pragma unroll
    for (int j = 0; j < NV; ++j) unsafeAtomicAdd(dst + (int64_t)key * stride + j, v[j]);
  }
}

// |.|max bookkeeping from many workgroups: same-address atomics serialise in L2 (~130 ns each), so look first -- a
// stale (lower) value only costs an atomic that was not needed
__device__ __forceinline__ void amax_publish(unsigned* slot, float v) {
  const unsigned bits = __float_as_uint(v);
  if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

// dst[i] = sum over replicas of src[r*stride + i], i < n  (cooperative; caller synchronises afterwards)
__device__ __forceinline__ void stat_gather(const double* src, int n, int stride, double* dst) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < kStatRep; ++r) s += src[r * stride + i];
    dst[i] = s;
  }
}

// two blocks in ONE round trip: dst[0..nA) from A, dst[nA..nA+nB) from B (different lanes take them; called back to
// back, two stat_gather loops are two dependent-latency episodes in every workgroup's prologue)
__device__ __forceinline__ void stat_gather2(const double* srcA, int nA, int strideA, const double* srcB, int nB,
                                             int strideB, double* dst) {
  for (int i = threadIdx.x; i < nA + nB; i += blockDim.x) {
    const double* src = i < nA ? srcA + i : srcB + (i - nA);
    const int stride = i < nA ? strideA : strideB;
    double s = 0;
#pragma unroll
    for (int r = 0; r < kStatRep; ++r) s += src[r * stride];
    dst[i] = s;
  }
}

// Wave totals of NV per-lane doubles, written to out[0 .. NV) by the lanes that end up holding them.  Round 5: the block
// epilogues of the few-edge passes were 2100-3400 cycles (profiles/r05_edge_stamps.txt), most of it one six-step DPP sum per
// value (two v_mov_dpp + one v_add_f64 per step: 18 instructions per value, 252 for pass B's 14).  Here pairs of values are
// folded across the wave's halves (v_permlane32_swap on both dwords + one add leaves value 2i in lanes 0-31 and 2i+1 in
// lanes 32-63), pairs of those across the 16-lane rows (v_permlane16_swap), and only the last four steps run inside a row:
// 3 + 1.5 + 12 / 4 = 5.25 instructions per value instead of 18.  Same additions in a different association: fp64.
__device__ __forceinline__ double swap_add_f64_32(double a, double b) {     // lanes 0-31: a[l] + a[l+32]; lanes 32-63: b[l-32] + b[l]
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add_f64_16(double a, double b) {     // rows 0 / 2: a's pair of rows, rows 1 / 3: b's
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int NV>
__device__ __forceinline__ void wave_sums_f64(const double (&v)[NV], double* out) {
  constexpr int N4 = (NV + 3) / 4;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < N4; ++q) {
    const double v0 = v[4 * q], v1 = 4 * q + 1 < NV ? v[4 * q + 1] : 0.0, v2 = 4 * q + 2 < NV ? v[4 * q + 2] : 0.0,
                 v3 = 4 * q + 3 < NV ? v[4 * q + 3] : 0.0;
    double u = swap_add_f64_16(swap_add_f64_32(v0, v1), swap_add_f64_32(v2, v3));   // rows 0..3: values 0, 2, 1, 3 of the quad
    u = dpp_add_f64<0x111, 0xf>(u);   // row_shr:1
    u = dpp_add_f64<0x112, 0xf>(u);   // row_shr:2
    u = dpp_add_f64<0x114, 0xf>(u);   // row_shr:4
    u = dpp_add_f64<0x118, 0xf>(u);   // row_shr:8: the row's last lane holds its total
    const int idx = 4 * q + wave_sum4_slot(lane);
    if ((lane & 15) == 15 && idx < NV) out[idx] = u;
  }
}

// Block-wide sum of NV per-thread doubles; result atomically added to this block's replica of dst.
// smem: at least NV * (blockDim.x/64) doubles.  All threads must call.
template <int NV>
__device__ __forceinline__ void block_atomic_add(const double (&v)[NV], double* dst_base, int stride, double* smem,
                                                 int block = -1) {   // block: replica chooser (default: blockIdx.x)
  double* dst = dst_base + ((block < 0 ? (int)blockIdx.x : block) % kStatRep) * stride;
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  wave_sums_f64<NV>(v, smem + wid * NV);
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += blockDim.x) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s += smem[w * NV + i];
    unsafeAtomicAdd(dst + i, s);
  }
  __syncthreads();
}

// BatchNorm (batch statistics) as y = s*z + t from fp64 (sum, sumsq) over `count` rows.
// Latency matters (this sits in kernel prologues): no fp64 division or square root -- multiply by 1/count,
// take v_rsq_f32 of the variance and polish it with one fp64 Newton step (error ~1e-14 relative).
__device__ __forceinline__ void bn_affine(double sum, double sumsq, double count, float gamma, float beta,
                                          float& s, float& t) {
  const double inv = 1.0 / count;            // count is a kernel argument: uniform, hoisted by the compiler
  const double mean = sum * inv;
  double var = fma(sumsq, inv, -mean * mean);
  var = (var < 0 ? 0 : var) + MTMC_BN_EPS;
  double r = (double)rsqrtf((float)var);
  r = r * fma(-0.5 * var, r * r, 1.5);
  const double sd = (double)gamma * r;
  s = (float)sd;
  t = (float)fma(-mean, sd, (double)beta);
}

// In-kernel stamps of the few-edge round kernels (-DEK_STAMP=1, tools/edge_stamps.py; tools/build_variant.sh): s_memtime at up to
// eight points of workgroup 0 and of the workgroup in the middle of the grid, thread 0.  One buffer per translation unit.
#ifndef EK_STAMP
#define EK_STAMP 0
#endif
#if EK_STAMP
constexpr int kEkKernels = 8, kEkPoints = 8;
static __device__ unsigned long long g_ek[kEkKernels * 2 * kEkPoints];
#define EK_T(kid, pt)                                                                                          \
  do {                                                                                                         \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))                                  \
      g_ek[((kid) * 2 + (blockIdx.x != 0)) * kEkPoints + (pt)] = __builtin_amdgcn_s_memtime();                 \
  } while (0)
#else
#define EK_T(kid, pt) do {} while (0)
#endif

// Dropout (training mode only; reference models/mlp.py:21-22): counter-based, so any kernel -- forward or
// backward, or the CPU model in tests/ -- regenerates the same mask from (seed, stream, element index).
// ONE 64-bit hash serves FOUR consecutive elements (round 5: a hash per element was most of what Dropout cost -- 30-odd
// instructions of 64-bit integer arithmetic for every activation): element idx belongs to group idx >> 2 and reads 16-bit field
// idx & 3 of the group's hash; keep iff field >= thresh, thresh = p * 2^16; kept values are scaled by 1/(1-p).  Kernels whose
// lanes hold four consecutive elements (drop_keep4 / drop_apply4) hash once per four; the element-wise calls agree with them.
// The node update's messages [E][32] are indexed CHANNEL-major for this (drop_msg_index): their kernels keep one channel per
// lane and walk along the edges, so four consecutive edges of a channel share a hash.
// on == 2 (MTMC_F_SEED_ON_DEVICE): `seed` is the ADDRESS of the seed word of this forward in the workspace / tape (written by
// seed_tick_kernel at the start of the forward from the caller's device counter), so that a HIP graph that holds the whole
// training step draws new masks on every replay; the backward of the same tape reads the same word.
struct Drop { unsigned long long seed; unsigned thresh; float inv_keep; int on; };
// Every kernel that draws masks calls drop_resolve on its own copy of the parameter ONCE, at its top (read inside drop_keep /
// drop_apply -- per element -- the indirection cost the hashing kernels 25 us of a 600 us training step,
// profiles/r05_seed_ab.txt); after it on is 0 or 1 and seed a value.
__device__ __forceinline__ void drop_resolve(Drop& d) {
  if (d.on == 2) {
    d.seed = *reinterpret_cast<const unsigned long long*>(d.seed);      // (uniform: a scalar load)
    d.on = 1;
  }
}

__host__ __device__ __forceinline__ unsigned long long drop_hash4(unsigned long long seed, unsigned stream, unsigned long long group) {
  unsigned long long z = group + seed * 0x9E3779B97F4A7C15ull + (unsigned long long)stream * 0xBF58476D1CE4E5B9ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ bool drop_field(const Drop& d, unsigned long long z, int j) {     // j = element index & 3
  return ((unsigned)(z >> (16 * j)) & 0xffffu) >= d.thresh;
}
__device__ __forceinline__ bool drop_keep(const Drop& d, unsigned stream, unsigned long long idx) {
  return !d.on || drop_field(d, drop_hash4(d.seed, stream, idx >> 2), (int)(idx & 3));
}
__device__ __forceinline__ float drop_apply(const Drop& d, unsigned stream, unsigned long long idx, float v) {
  if (!d.on) return v;
  return drop_field(d, drop_hash4(d.seed, stream, idx >> 2), (int)(idx & 3)) ? v * d.inv_keep : 0.f;
}
// four consecutive elements idx .. idx + 3 (idx % 4 == 0: one hash; anything else: element by element)
__device__ __forceinline__ void drop_apply4(const Drop& d, unsigned stream, unsigned long long idx, float (&v)[4]) {
  if (!d.on) return;
  if ((idx & 3) == 0) {
    const unsigned long long z = drop_hash4(d.seed, stream, idx >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // (one v_mul_f32 each, kept apart: paired up by the SLP vectoriser they become v_pk_mul_f32 with op_sel on the scalar
      //  pair that holds inv_keep -- the form tools/check_isa.py bans from kernels with MFMAs)
      float r = v[j] * d.inv_keep;
      asm volatile("" : "+v"(r));
      v[j] = drop_field(d, z, j) ? r : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = drop_apply(d, stream, idx + j, v[j]);
  }
}
// element index of message (edge e, channel k) of the node update: channel-major over the edge count padded to 4
__host__ __device__ __forceinline__ unsigned long long drop_msg_index(long long e, int k, long long n_edges) {
  return (unsigned long long)k * (unsigned long long)((n_edges + 3) & ~3ll) + (unsigned long long)e;
}
// stream ids
constexpr unsigned kDropEncEdge1 = 1, kDropEncEdge2 = 2, kDropEncNode = 100, kDropRound = 1000;   // +layer / +2r(+1)

// Small per-launch parameter blocks, read through uniform (scalar) loads.
struct EdgeEncParams {      // encoder.edge_mlp: in(1|2) -> 4 -> 4
  const float* w1; const float* b1; const float* g1; const float* bt1;
  const float* w2; const float* b2; const float* g2; const float* bt2;
  const double* stat_attr;  // f64[kStatRep][kAttrStride]
  const double* stat_enc2;  // f64[kStatRep][kEnc2Stride]
  float* aff;               // f32[16] = EdgeEncAffine, finalised once per forward by node_proj_kernel (round 0)
  int fe;                   // edge_in_dim (1 or 2)
  Drop drop;                // encoder dropout (training)
};

// Affine coefficients of the two edge-encoder BatchNorms, derived from moments:
//   layer 1 pre-activation z = W1 a + b1 : E[z_k] = W1_k.m1 + b1_k, E[z_k^2] = W1_k M2 W1_k^T + 2 b1_k W1_k.m1 + b1_k^2
struct EdgeEncAffine { float s1[4], t1[4], s2[4], t2[4]; };

// quadratic form w^T M w for a packed symmetric M
__device__ __forceinline__ double quad_form(const float* w, int n, const double* m2) {
  double q = 0;
  for (int i = 0; i < n; ++i) {
    q += (double)w[i] * (double)w[i] * m2[tri(n, i, i)];
    for (int j = i + 1; j < n; ++j) q += 2.0 * (double)w[i] * (double)w[j] * m2[tri(n, i, j)];
  }
  return q;
}

__device__ __forceinline__ void moments_affine(const float* w, int in_dim, float bias, const double* m1,
                                               const double* m2, double count, float gamma, float beta,
                                               float& s, float& t) {
  double wm1 = 0;
  for (int i = 0; i < in_dim; ++i) wm1 += (double)w[i] * m1[i];
  const double q = quad_form(w, in_dim, m2);
  const double b = bias;
  const double sum = wm1 + b * count;                       // sum over rows of z
  const double sumsq = q + 2.0 * b * wm1 + b * b * count;   // sum over rows of z^2
  bn_affine(sum, sumsq, count, gamma, beta, s, t);
}

// Whole block: add up the replicated moments (scratch: kStatAttr + kStatEnc2 doubles of shared memory), then
// threads 0..3 derive the affines.  `which`: 1 = layer-1 only, 2 = both.  Ends with a barrier.
__device__ __forceinline__ void edge_enc_affine_to_smem(const EdgeEncParams& p, double count, int which,
                                                        EdgeEncAffine* out, double* scratch) {
  stat_gather(p.stat_attr, kStatAttr, kAttrStride, scratch);
  if (which >= 2) stat_gather(p.stat_enc2, kStatEnc2, kEnc2Stride, scratch + kStatAttr);
  __syncthreads();
  const int k = threadIdx.x;
  if (k < 4) {
    moments_affine(p.w1 + k * p.fe, p.fe, p.b1[k], scratch, scratch + 2, count, p.g1[k], p.bt1[k],
                   out->s1[k], out->t1[k]);
    if (which >= 2)
      moments_affine(p.w2 + k * 4, 4, p.b2[k], scratch + kStatAttr, scratch + kStatAttr + 4, count, p.g2[k], p.bt2[k],
                     out->s2[k], out->t2[k]);
  }
  __syncthreads();
}

// Passes A/B: the affines were finalised once (EdgeEncParams::aff); fetch the 16 floats.  Ends with a barrier.
__device__ __forceinline__ void edge_enc_affine_load(const EdgeEncParams& p, EdgeEncAffine* out) {
  if (threadIdx.x < 16) reinterpret_cast<float*>(out)[threadIdx.x] = p.aff[threadIdx.x];
  __syncthreads();
}

// u = relu(bn1(W1 a + b1)) : hidden layer of the edge encoder
__device__ __forceinline__ void edge_enc_hidden(const EdgeEncParams& p, const EdgeEncAffine& af, int64_t e, float a0,
                                                float a1, float (&u)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float z = p.b1[k] + p.w1[k * p.fe] * a0;
    if (p.fe > 1) z = fmaf(p.w1[k * p.fe + 1], a1, z);
    u[k] = fmaxf(fmaf(z, af.s1[k], af.t1[k]), 0.f);
  }
  drop_apply4(p.drop, kDropEncEdge1, (unsigned long long)e * 4, u);      // the edge's four channels: one hash
}

// e0 = relu(bn2(W2 u + b2)) : output of the edge encoder
__device__ __forceinline__ void edge_enc_out(const EdgeEncParams& p, const EdgeEncAffine& af, int64_t eidx,
                                             const float (&u)[4], float (&e)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float z = p.b2[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) z = fmaf(p.w2[k * 4 + j], u[j], z);
    e[k] = fmaxf(fmaf(z, af.s2[k], af.t2[k]), 0.f);
  }
  drop_apply4(p.drop, kDropEncEdge2, (unsigned long long)eidx * 4, e);
}

__device__ __forceinline__ void load_attr(const float* attr, int fe, int64_t e, float& a0, float& a1);

// Moments of the edge encoder's hidden activations (enc2): workgroup `block` of `n_blocks`, 256 threads.  The body of
// enc2_kernel (edge_kernels.hip) and of the passenger workgroups the few-row encoder GEMM carries (gemm_bn.hip).
__device__ __forceinline__ void enc2_body(const EdgeEncParams& enc_in, const float* attr, int64_t n_edges, double e_total,
                                          double* stat_enc2, int block, int n_blocks) {
  EdgeEncParams enc = enc_in;
  drop_resolve(enc.drop);
  __shared__ EdgeEncAffine af;
  __shared__ double red[14 * 4];
  edge_enc_affine_to_smem(enc, e_total, 1, &af, red);
  double acc[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] = 0;
  const int64_t nthreads = (int64_t)n_blocks * 256;
  for (int64_t e = (int64_t)block * 256 + threadIdx.x; e < n_edges; e += nthreads) {
    float a0, a1, u[4];
    load_attr(attr, enc.fe, e, a0, a1);
    edge_enc_hidden(enc, af, e, a0, a1, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += u[i];
#pragma unroll
      for (int j = i; j < 4; ++j) acc[4 + tri(4, i, j)] += (double)u[i] * u[j];
    }
  }
  block_atomic_add<14>(acc, stat_enc2, kEnc2Stride, red, block);
}

// the same for a stream's last reader (non-temporal: the line is not kept in L2 / Infinity Cache)
__device__ __forceinline__ void load_attr_nt(const float* attr, int fe, int64_t e, float& a0, float& a1) {
  if (fe == 2) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(attr) + e);
    a0 = v[0]; a1 = v[1];
  } else {
    a0 = __builtin_nontemporal_load(attr + e); a1 = 0.f;
  }
}

__device__ __forceinline__ void load_attr(const float* attr, int fe, int64_t e, float& a0, float& a1) {
  if (fe == 2) {
    const float2 v = reinterpret_cast<const float2*>(attr)[e];
    a0 = v.x; a1 = v.y;
  } else {
    a0 = attr[e]; a1 = 0.f;
  }
}

}  // namespace mtmc
