// Per-edge passes of the message-passing network on gfx950 (MI355X).
//
// These are the HBM-bound kernels: each streams the edge list once, gathers 16-byte per-node
// projections instead of the reference's 128-byte node rows (W.[h[row]|h[col]|e] = Pr[row]+Pc[col]+We.e),
// and never materialises the [E,68] / [E,36] / [E,32] intermediates of the reference
// (reference models/mpn.py:68, :97-98).  BatchNorm batch statistics are accumulated in fp64.
//
//   prep_kernel        int64 strided edge_index -> int32 row/col, out-degree, moments of edge_attr
//   enc2_kernel        moments of the edge-encoder hidden activations
//   pass_a_kernel      statistics of z1 = We.[h[row]|h[col]|e] + be               (EdgeModel, mpn.py:67-69)
//   pass_b_kernel      e' = relu(bn(z1)) stored; moments of e'; per-node segment sums of e'
//   pass_c_kernel      m = relu(bn(Wn.[h[row]|e'] + bn)); h' = agg_row(m); logits  (NodeModel, mpn.py:97-99)
//   classify_e0_kernel logits of the encoded edges when L == 0                      (mpn.py:295-297)
#include "kernels.h"
#include "prep_body.h"
#include "split_body.h"

#include <type_traits>

namespace mtmc {

// ------------------------------------------------------------------------------------------------
// prep
// ------------------------------------------------------------------------------------------------
template <bool SPLIT>      // SPLIT: the kernel also carries operand-split jobs (64 more registers per lane: few-edge graphs only)
__device__ void amax_jobs(const PrepParams& p, int block) {
  __shared__ float wmax[4];
  int j = 0;
  while (j + 1 < p.n_jobs && block >= p.jobs[j + 1].block0) ++j;      // every passenger workgroup serves ONE job
  const AmaxJob job = p.jobs[j];
  if (SPLIT && job.kind == kJobSplit) {        // operand split (x planes of few-row graphs / the weight-plane cache): split_body.h
    __shared__ unsigned long long fp_red[4];
    split_rows_body(job.ptr, job.ld, job.rows, job.cols, job.planes, job.rows * (int64_t)job.cols, job.inv, 0, job.rows,
                    block - job.block0, job.fp, fp_red, job.out);
    return;
  }
  const int c4n = job.cols / 4;                         // cols is a multiple of 32 (check_model)
  const int64_t total = job.rows * c4n;
  const bool dense = job.ld == job.cols;                // weights and contiguous x: no row arithmetic
  float m = 0.f;
  const int64_t stride = (int64_t)job.n_blocks * 256;
  auto at = [&](int64_t i) {
    const float* src = dense ? job.ptr + i * 4 : job.ptr + (i / c4n) * job.ld + (i % c4n) * 4;
    return *reinterpret_cast<const float4*>(src);
  };
  auto fold = [&](const float4 v) { m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  int64_t i = (int64_t)(block - job.block0) * 256 + threadIdx.x;
  for (; i + 3 * stride < total; i += 4 * stride) {       // four independent 16-byte loads in flight per lane
    const float4 v0 = at(i), v1 = at(i + stride), v2 = at(i + 2 * stride), v3 = at(i + 3 * stride);
    fold(v0); fold(v1); fold(v2); fold(v3);
  }
  for (; i < total; i += stride) fold(at(i));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (b > 0.f) atomicMax(job.out + (block % kAmaxRep), __float_as_uint(b));
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void prep_kernel(PrepParams p) {
  // passenger workgroups (operand scales / operand splits of the node encoder, the weight-plane cache's verification) come
  // FIRST in the grid: they are the longest (64 KB of reads each) and nothing else of the forward can start before the last
  // of them is done
  if ((int)blockIdx.x < p.n_pass_blocks) {
    amax_jobs<SPLIT>(p, blockIdx.x);
    return;
  }
  prep_edge_body<4>(p, (int)blockIdx.x - p.n_pass_blocks);      // (prep_body.h)
}

// The operand-split jobs by themselves (many-edge graphs: inside prep_kernel their 64 registers per lane cost the edge loop
// three waves per SIMD of occupancy -- 67 -> 109 us at config 4, profiles/r05_cfg4_kernel_stats.csv before / after)
__global__ __launch_bounds__(256) void split_jobs_kernel(PrepParams p) { amax_jobs<true>(p, blockIdx.x); }

// MTMC_F_SEED_ON_DEVICE: this forward's Dropout seed = the caller's device counter, which moves on by one
__global__ void seed_tick_kernel(unsigned long long* counter, unsigned long long* word) {
  const unsigned long long v = *counter;
  *word = v;
  *counter = v + 1;
}

// ------------------------------------------------------------------------------------------------
// edge encoder, hidden-layer moments
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc2_kernel(EdgeEncParams enc, const float* attr, int64_t n_edges,
                                                   double e_total, double* stat_enc2) {
  enc2_body(enc, attr, n_edges, e_total, stat_enc2, blockIdx.x, gridDim.x);   // (common.h: also carried by the few-row GEMM)
}

// ------------------------------------------------------------------------------------------------
// z1 of one edge: Pr[row] + Pc[col] + We_e . e_in + be
// ------------------------------------------------------------------------------------------------
// The small constants every edge needs in pass A -- the two edge-encoder layers with their BatchNorm affines (first round /
// reattached edges: e0 is recomputed from the 8-byte attributes, never stored) and the edge-update weights: 68 floats in
// the first round, 84 with reattached edges.  As uniform (scalar) operands they did not all fit the SGPR file (105-160
// spills, each a v_readlane in the loop: the first round's launch took 220 us at config 4 against 133 us for the later
// rounds, which move MORE bytes); all in vector registers they cost two waves per SIMD of occupancy.  So they are SPLIT:
// the hidden layer (20 floats) and, without reattached edges, the 4 x 4 update weights (20) stay scalar -- read through
// uniform pointers --, the output layer (28) and the 4 x 8 update weights of the reattached forms (36) are staged once per
// workgroup in LDS and read back by every lane: a load from LDS lands in a VGPR and stays there.
// Same arithmetic, same order as edge_enc_hidden / edge_enc_out (common.h).
struct EdgeConstsV {                       // the LDS-staged (vector-register) part
  float w2[4][4], b2[4], s2[4], t2[4];     // output layer of the edge encoder
  float uw[4][8], ub[4];                   // reattached forms only: [W_e0 | W_e] columns of the edge update, bias
};
constexpr int kEdgeConstsV = sizeof(EdgeConstsV) / sizeof(float);
struct EdgeConstsS {                       // the scalar part: copied out of memory ONCE, before the edge loop -- read through
  float w1[4][2], b1[4], s1[4], t1[4];     // the parameter pointers inside the loop they would be re-fetched after every z1
  float uw[4][4], ub[4];                   // store (which may alias them as far as the compiler knows)
};
template <int MODE>
__device__ __forceinline__ void load_edge_consts_s(const RoundParams& p, EdgeConstsS& k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k.w1[i][0] = MODE != 0 ? p.enc.w1[i * p.enc.fe] : 0.f;
    k.w1[i][1] = (MODE != 0 && p.enc.fe > 1) ? p.enc.w1[i * p.enc.fe + 1] : 0.f;
    k.b1[i] = MODE != 0 ? p.enc.b1[i] : 0.f;
    k.s1[i] = MODE != 0 ? p.enc.aff[i] : 0.f;            // p.enc.aff = s1[4] | t1[4] | s2[4] | t2[4] (EdgeEncAffine)
    k.t1[i] = MODE != 0 ? p.enc.aff[4 + i] : 0.f;
    k.ub[i] = (MODE & 2) ? 0.f : p.ue_b[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) k.uw[i][j] = (MODE & 2) ? 0.f : p.ue_w[i * p.ue_ld + p.ue_eoff + j];
  }
}

// whole block; ends with a barrier.  MODE as in pass_a_kernel.
template <int MODE>
__device__ __forceinline__ void stage_edge_consts(const RoundParams& p, EdgeConstsV* cs) {
  float* dst = reinterpret_cast<float*>(cs);
  if (MODE != 0) {
    for (int i = threadIdx.x; i < kEdgeConstsV; i += blockDim.x) {
      float v;
      if (i < 16) v = p.enc.w2[i];
      else if (i < 20) v = p.enc.b2[i - 16];
      else if (i < 28) v = p.enc.aff[8 + i - 20];                                  // s2[4] | t2[4] of EdgeEncAffine
      else if (i < 60) v = (MODE & 2) ? p.ue_w[((i - 28) >> 3) * p.ue_ld + p.ue_eoff + ((i - 28) & 7)] : 0.f;
      else v = p.ue_b[i - 60];
      dst[i] = v;
    }
  }
  __syncthreads();
}

struct PrevAffine { float s[4], t[4]; };   // BatchNorm affine of the previous round's z1 (lazy e')

// z1 of one edge: Pr[row] + Pc[col] + We_e . e_in + be.
// MODE 0: a later round without reattached edges (no attribute loads, no edge-encoder arithmetic, 4 x 4 weights);
// bit 0: first round, bit 1: reattach_initial_edges.
// Loads and arithmetic are separate steps so that a thread's kEPT edges have ALL their loads in flight before the first
// result is stored (the z1 store may alias every input as far as the compiler can tell: interleaved, each edge's load
// chain would start only after the previous edge's store).
struct EdgeIn { float4 pr, pc, ev; float a0, a1; };
// Non-temporal access on the edge streams (tools/edge_nt_sweep.sh; config 4 / config 5, all edge passes of a forward, us):
//   none 737 / 8916 | PA 3 698 / 8892 | PA 4 717 / 8759 | PA 7 708 / 8651 | PA 3 + PC 735 / 8876 | PA 3 + PB 684 / 8698 |
//   PA 7 + PC + PB 724 / 8487.  With PA 11 + PB: 675 / 8714; PA 15 + PB: 689 / 8510 (hence RoundParams::stream_z1).
// The streams of a 10M-edge list (z1: 160 MB) survive in the 256 MB Infinity Cache from one pass to the next unless a pass
// in between pushes them out; those of a 100M-edge list never do.  Taken: row / col ids and the previous z1 non-temporal
// in pass A (its L2 keeps the randomly gathered Pc table instead: 117 -> 99 us per launch at config 4), pass B's stream
// loads non-temporal (48 -> 40 us), z1 stored normally and read normally by pass C so that the next pass finds it cached --
// the regime of config 4 and of an eighth of config 5.
#ifndef PA_NT
#define PA_NT 11          // pass A: 1 row / col ids, 2 previous z1, 4 the z1 store, 8 edge_attr in the first round (126 -> 120 us)
#endif
#ifndef PC_NT
#define PC_NT 0           // stream loads of pass_c_sorted_kernel
#endif
#ifndef PB_NT
#define PB_NT 1           // stream loads of pass B (whole tiles)
#endif

template <int MODE>
__device__ __forceinline__ void edge_load_rc(const RoundParams& p, int64_t e, int r, int col, EdgeIn& in);
template <int MODE>
__device__ __forceinline__ void edge_load(const RoundParams& p, int64_t e, EdgeIn& in) {
#if PA_NT & 1
  const int r = __builtin_nontemporal_load(p.row32 + e), col = __builtin_nontemporal_load(p.col32 + e);
#else
  const int r = p.row32[e], col = p.col32[e];
#endif
  edge_load_rc<MODE>(p, e, r, col, in);
}
// ... with the row / column ids in hand (the column-blocked traversal knows the row without loading it)
template <int MODE>
__device__ __forceinline__ void edge_load_rc(const RoundParams& p, int64_t e, int r, int col, EdgeIn& in) {
  // P = [Pr: N x 4 | Pc: N x 4]: the randomly gathered half is a compact 16 B/node table (four nodes per 64-byte sector)
  in.pr = *reinterpret_cast<const float4*>(p.P + (int64_t)r * 4);
  in.pc = *reinterpret_cast<const float4*>(p.P + ((int64_t)p.n_nodes + col) * 4);
  in.a0 = in.a1 = 0.f;
  in.ev = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1 && (PA_NT & 8)) load_attr_nt(p.attr, p.enc.fe, e, in.a0, in.a1);   // first round, no reattachment: the last reader
  else if (MODE != 0) load_attr(p.attr, p.enc.fe, e, in.a0, in.a1);
#if PA_NT & 2
  if (!(MODE & 1)) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p.e_prev) + e);
    in.ev = make_float4(v[0], v[1], v[2], v[3]);
  }
#else
  if (!(MODE & 1)) in.ev = reinterpret_cast<const float4*>(p.e_prev)[e];
#endif
}

template <int MODE, bool DROP>
__device__ __forceinline__ void edge_z1(const RoundParams& p, const EdgeConstsS& ks, const EdgeConstsV& c,
                                        const PrevAffine& pa, int64_t e, const EdgeIn& in, float (&z)[4]) {
  constexpr bool first_round = (MODE & 1) != 0, reattach = (MODE & 2) != 0;
  float e0[4] = {0, 0, 0, 0}, ep[4];
  if (first_round || reattach) {
    float u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // edge_enc_hidden, scalar operands (edge_in_dim 1: w1[k][1] = a1 = 0)
      const float zz = fmaf(ks.w1[k][1], in.a1, ks.b1[k] + ks.w1[k][0] * in.a0);
      u[k] = fmaxf(fmaf(zz, ks.s1[k], ks.t1[k]), 0.f);
    }
    if (DROP) drop_apply4(p.enc.drop, kDropEncEdge1, (unsigned long long)e * 4, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // edge_enc_out
      float zz = c.b2[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) zz = fmaf(c.w2[k][j], u[j], zz);
      e0[k] = fmaxf(fmaf(zz, c.s2[k], c.t2[k]), 0.f);
    }
    if (DROP) drop_apply4(p.enc.drop, kDropEncEdge2, (unsigned long long)e * 4, e0);
  }
  if (first_round) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ep[j] = e0[j];
  } else {
    ep[0] = in.ev.x; ep[1] = in.ev.y; ep[2] = in.ev.z; ep[3] = in.ev.w;
    if (p.lazy_e) {                               // the buffer holds the previous round's z1: e' = relu(bn(z1))
#pragma unroll
      for (int j = 0; j < 4; ++j) ep[j] = fmaxf(fmaf(ep[j], pa.s[j], pa.t[j]), 0.f);
    }
  }
  const float prv[4] = {in.pr.x, in.pr.y, in.pr.z, in.pr.w}, pcv[4] = {in.pc.x, in.pc.y, in.pc.z, in.pc.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (reattach) {
      float acc = prv[k] + pcv[k] + c.ub[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fmaf(c.uw[k][j], e0[j], acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fmaf(c.uw[k][4 + j], ep[j], acc);
      z[k] = acc;
    } else {                                            // 4 x 4 weights + bias: scalar operands
      float acc = prv[k] + pcv[k] + ks.ub[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fmaf(ks.uw[k][j], ep[j], acc);
      z[k] = acc;
    }
  }
}

// kEPT = edges per thread and loop trip in passes A/B: four independent load chains in flight on big graphs, one on
// small ones, where filling the 256 CUs with waves matters more (pick_ept).  DROP: the edge encoder's Dropout is compiled
// in (training); eval-mode kernels carry neither its hash arithmetic nor its scalar operands.

template <int kEPT, int MODE, bool DROP>
__global__ __launch_bounds__(256, (MODE == 0 ? 5 : 1)) void pass_a_kernel(RoundParams p) {   // plain later round: <= 96 VGPRs
  if (DROP) drop_resolve(p.enc.drop);
  __shared__ EdgeConstsV cs_s;
  __shared__ double red[8 * 4];
  __shared__ PrevAffine pa_s;
  EdgeConstsS ks;                                  // first thing in the kernel: ahead of every store and barrier these
  load_edge_consts_s<MODE>(p, ks);                 // uniform loads are scalar loads (SGPRs); behind one they become vector loads
  if (p.col_blocks > 0 && p.flags[0] == 0 && p.flags[2] == 0) return;   // pass_a_blocked_kernel, launched just before, did this round
  EK_T(MODE == 0 ? 1 : 2, 0);
  if (p.lazy_e && !(MODE & 1)) {
    stat_gather(p.prev_stats + kRoundZ1Off, 8, kZ1Stride, red);
    __syncthreads();
    if (threadIdx.x < 4)
      bn_affine(red[threadIdx.x], red[4 + threadIdx.x], p.e_total, p.ue_g[threadIdx.x], p.ue_bt[threadIdx.x],
                pa_s.s[threadIdx.x], pa_s.t[threadIdx.x]);
  }
  stage_edge_consts<MODE>(p, &cs_s);               // (barrier inside: pa_s is visible too)
  EK_T(MODE == 0 ? 1 : 2, 1);
  PrevAffine pa;
#pragma unroll
  for (int j = 0; j < 4; ++j) { pa.s[j] = pa_s.s[j]; pa.t[j] = pa_s.t[j]; }
  EdgeConstsV c;                                   // per-lane copy: VGPRs (only the fields MODE uses survive)
  {
    const float* src = reinterpret_cast<const float*>(&cs_s);
    float* dst = reinterpret_cast<float*>(&c);
#pragma unroll
    for (int i = 0; i < kEdgeConstsV; ++i) dst[i] = MODE != 0 ? src[i] : 0.f;
  }
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto finish = [&](int64_t e, const EdgeIn& in) {
    float z[4];
    edge_z1<MODE, DROP>(p, ks, c, pa, e, in, z);
    // the random 16-byte P[col] gather is what bounds this pass (one cache line per lane): do it once and
    // hand z1 to pass B through memory instead of gathering again there
    if ((PA_NT & 4) || p.stream_z1) {                 // (block-uniform)
      typedef float f4v __attribute__((ext_vector_type(4)));
      f4v zv = {z[0], z[1], z[2], z[3]};
      __builtin_nontemporal_store(zv, reinterpret_cast<f4v*>(p.e_buf) + e);
    } else {
      reinterpret_cast<float4*>(p.e_buf)[e] = make_float4(z[0], z[1], z[2], z[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] += z[k];
      acc[4 + k] += (double)z[k] * z[k];
    }
  };
  // whole tiles of 256 * kEPT edges without a bounds check per edge (no exec-mask bookkeeping in the hot loop) ...
  constexpr int64_t kTile = 256 * kEPT;
  const int64_t n_full = p.n_edges / kTile;
  for (int64_t t = blockIdx.x; t < n_full; t += gridDim.x) {
    const int64_t base = t * kTile + threadIdx.x;
    EdgeIn in[kEPT];
#pragma unroll
    for (int i = 0; i < kEPT; ++i) edge_load<MODE>(p, base + i * 256, in[i]);
#pragma unroll
    for (int i = 0; i < kEPT; ++i) finish(base + i * 256, in[i]);
  }
  // ... and the last, partial tile: one 256-edge piece per block
  for (int i = blockIdx.x; i < kEPT; i += gridDim.x) {
    const int64_t e = n_full * kTile + (int64_t)i * 256 + threadIdx.x;
    if (e < p.n_edges) {
      EdgeIn in;
      edge_load<MODE>(p, e, in);
      finish(e, in);
    }
  }
  EK_T(MODE == 0 ? 1 : 2, 2);
  block_atomic_add<8>(acc, p.stats + kRoundZ1Off, kZ1Stride, red);
  EK_T(MODE == 0 ? 1 : 2, 3);
}

// ------------------------------------------------------------------------------------------------
// Pass A by COLUMN BLOCKS (graphs whose Pc table outgrows an XCD's 4 MB L2: config 5, N = 1M -> 16 MB).
// In edge order every 16-byte Pc[col] gather of such a graph misses the L2 and fetches a 64-byte sector of its own from the
// Infinity Cache: 6.1 GB fetched for 2.4 GB of algorithmic reads per launch, 1.85 ms.  A row-sorted list
// with ascending columns inside a row (the reference's lists are: inference.py:407-413 builds them as cartesian products
// of ascending node lists) is also sorted by column INSIDE every row, so the edges of row i whose column falls into block b
// are one contiguous sub-run [sub[i][b], sub[i][b+1]).  colblock_index_kernel finds the B - 1 inner boundaries of every row
// once per forward (binary searches inside the row's own column segment); pass_a_blocked_kernel then walks
// (256-row chunk) x (column block) pieces, the block bound to blockIdx % 8 -- under round-robin placement one XCD, whose L2
// then serves all gathers from a 2 MB slice of Pc (placement changes speed only).  Per wave: 64 rows' sub-run lengths ->
// inclusive scan -> every lane takes slots k, k + 64, ... of the wave's concatenated sub-runs and finds its row by a 6-step
// binary search over the scan in LDS.  z1 lands where it always does (edge order in memory is unchanged), so passes B / C and
// the next round are untouched.  prep_kernel's flags decide on the device: unsorted rows or columns -> this kernel returns
// and pass_a_kernel, launched behind it, does the round (and vice versa).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colblock_index_kernel(const int* __restrict__ col32, const int* __restrict__ row_start,
                                                             const int* __restrict__ deg, const int* __restrict__ flags,
                                                             int64_t row_lo, int64_t row_hi, int B, int blk_nodes,
                                                             int* __restrict__ sub) {
  if (flags[0] != 0 || flags[2] != 0) return;
  const int64_t n_items = (row_hi - row_lo) * (B + 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
    const int64_t r = row_lo + i / (B + 1);
    const int b = (int)(i % (B + 1));
    const int d = deg[r];
    int pos = 0;
    if (d > 0) {
      const int s0 = row_start[r];
      if (b == 0) pos = s0;
      else if (b == B) pos = s0 + d;
      else {                                                 // first edge of the row with col >= b * blk_nodes
        const int want = b * blk_nodes;
        int lo = s0, hi = s0 + d;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (col32[mid] < want) lo = mid + 1; else hi = mid;
        }
        // Inner boundaries snap DOWN to a multiple of four edges = one 64-byte sector of the 16-byte-per-edge streams (z1 in and
        // out): a sector is then read and written by ONE column block, i.e. one XCD -- unsnapped, the first and last sector of
        // every ~200-byte sub-run were shared with the neighbouring block on another XCD, fetched twice and written in two
        // partial pieces.  The <= 3 edges this moves into the neighbour's block gather from the neighbour's slice of Pc (a miss
        // in this XCD's L2, nothing else); monotone boundaries stay monotone under rounding.
        pos = lo & ~3;
        pos = pos < s0 ? s0 : pos;
      }
    }
    sub[(r - row_lo) * (B + 1) + b] = pos;                   // (a row without edges: all zeros -> empty sub-runs)
  }
}

template <int MODE>
__global__ __launch_bounds__(256, (MODE == 0 ? 4 : 1)) void pass_a_blocked_kernel(RoundParams p, const int* __restrict__ sub,
                                                                                  int B, int64_t row_lo, int64_t row_hi) {
  __shared__ EdgeConstsV cs_s;
  __shared__ double red[8 * 4];
  __shared__ PrevAffine pa_s;
  __shared__ int pre_s[4][64], base_s[4][64];
  EdgeConstsS ks;
  load_edge_consts_s<MODE>(p, ks);
  if (p.flags[0] != 0 || p.flags[2] != 0) return;   // unsorted rows / columns: pass_a_kernel does this round (block-uniform)
  if (p.lazy_e && !(MODE & 1)) {
    stat_gather(p.prev_stats + kRoundZ1Off, 8, kZ1Stride, red);
    __syncthreads();
    if (threadIdx.x < 4)
      bn_affine(red[threadIdx.x], red[4 + threadIdx.x], p.e_total, p.ue_g[threadIdx.x], p.ue_bt[threadIdx.x],
                pa_s.s[threadIdx.x], pa_s.t[threadIdx.x]);
  }
  stage_edge_consts<MODE>(p, &cs_s);
  PrevAffine pa;
#pragma unroll
  for (int j = 0; j < 4; ++j) { pa.s[j] = pa_s.s[j]; pa.t[j] = pa_s.t[j]; }
  EdgeConstsV c;
  {
    const float* src = reinterpret_cast<const float*>(&cs_s);
    float* dst = reinterpret_cast<float*>(&c);
#pragma unroll
    for (int i = 0; i < kEdgeConstsV; ++i) dst[i] = MODE != 0 ? src[i] : 0.f;
  }
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // column block of this workgroup: blockIdx % 8 picks the XCD-bound residue, the next bits the block among that XCD's B / 8
  const int per_xcd = B >> 3;
  const int b = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) % per_xcd);
  const int64_t group = (blockIdx.x >> 3) / per_xcd, n_groups = (gridDim.x >> 3) / per_xcd;
  const int64_t n_chunks = (row_hi - row_lo + 255) / 256;
  constexpr int U = 4;                              // slots per lane and trip: four independent load chains in flight
  // a lane's row of the chunk: its sub-run [s0, s0 + len) of column block b.  The NEXT chunk's pair is requested before this
  // chunk's edges are walked (a dependent global load at the head of every ~800-edge chunk would idle the wave for its latency)
  auto sub_of = [&](int64_t chunk, int& s0, int& len) {
    const int64_t r = row_lo + chunk * 256 + w * 64 + lane;
    s0 = 0; len = 0;
    if (chunk < n_chunks && r < row_hi) {
      const int* sb = sub + (r - row_lo) * (B + 1) + b;
      s0 = sb[0];
      len = sb[1] - s0;
    }
  };
  int s0_n, len_n;
  sub_of(group, s0_n, len_n);
  for (int64_t chunk = group; chunk < n_chunks; chunk += n_groups) {
    const int64_t r0 = row_lo + chunk * 256 + w * 64;          // this wave's 64 rows
    const int s0 = s0_n, len = len_n;
    sub_of(chunk + n_groups, s0_n, len_n);
    int incl = len;                                   // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    pre_s[w][lane] = incl;
    base_s[w][lane] = s0 - (incl - len);              // edge index of slot k inside this row's sub-run: base + k
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Branch-free up to the stores: slots past the wave's total are clamped to its last slot (loaded and computed again,
    // never stored or counted), so that the UU searches, then the UU column loads, then the UU x 3 data loads are each in
    // flight together -- behind a per-slot `if` every slot's chain (search -> column id -> gather) would run after the
    // previous one's.  Whole trips of U x 64 slots take UU = U; what is left takes single 64-slot trips (a wave's ~800 slots
    // are 3 whole trips + 32: one clamped trip of four would redo a fifth of the work).
    auto trip = [&](int k0, auto uu_tag) {
      constexpr int UU = decltype(uu_tag)::value;
      EdgeIn in[UU];
      int64_t ee[UU];
      int rw[UU], cl[UU];
      bool ok[UU];
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        int k = k0 + u * 64 + lane;
        ok[u] = k < total;
        k = ok[u] ? k : total - 1;
        int j = 0;                                    // first row of the wave whose inclusive count exceeds k
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (pre_s[w][j + step - 1] <= k) j += step;
        ee[u] = (int64_t)base_s[w][j] + k;
        rw[u] = (int)(r0 + j);
      }
#pragma unroll
      for (int u = 0; u < UU; ++u) {
#if PA_NT & 1
        cl[u] = __builtin_nontemporal_load(p.col32 + ee[u]);
#else
        cl[u] = p.col32[ee[u]];
#endif
      }
#pragma unroll
      for (int u = 0; u < UU; ++u) edge_load_rc<MODE>(p, ee[u], rw[u], cl[u], in[u]);
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        float z[4];
        edge_z1<MODE, false>(p, ks, c, pa, ee[u], in[u], z);
        if (ok[u]) {
          if ((PA_NT & 4) || p.stream_z1) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v zv = {z[0], z[1], z[2], z[3]};
            __builtin_nontemporal_store(zv, reinterpret_cast<f4v*>(p.e_buf) + ee[u]);
          } else {
            reinterpret_cast<float4*>(p.e_buf)[ee[u]] = make_float4(z[0], z[1], z[2], z[3]);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zk = ok[u] ? z[k] : 0.f;
          acc[k] += zk;
          acc[4 + k] += (double)zk * zk;
        }
      }
    };
    int k0 = 0;
    for (; k0 + 64 * U <= total; k0 += 64 * U) trip(k0, std::integral_constant<int, U>());
    for (; k0 < total; k0 += 64) trip(k0, std::integral_constant<int, 1>());
    __builtin_amdgcn_wave_barrier();                  // (the next chunk overwrites this wave's LDS rows)
  }
  block_atomic_add<8>(acc, p.stats + kRoundZ1Off, kZ1Stride, red);
}

// e' = relu(bn(z1)) in place; first / second moments of e'; and the edge-dependent part of the node-update statistics.
// z2 = Q[row] + A e' + b (32-wide) is never materialised for its BatchNorm statistics:
//   sum_e z2_k   = sum_i deg_i qb_ik                          + A_k . m1            qb = Q + b, m1 = sum_e e'
//   sum_e z2_k^2 = sum_i deg_i qb_ik^2 + 2 A_k . C[:,k]       + A_k M2 A_k^T        C[j][k] = sum_e e'_j qb[row_e][k]
// The node-only sums come from node_proj (it has Q in registers), M2 is this pass's second moments (the consumers add the
// quadratic form), and C -- 4 x 32 numbers -- is accumulated here PER RUN of equal rows: a run's four channel sums (the wave
// sums this pass takes anyway) times the row's qb (one 128-byte line), two fp64 FMAs per lane.  Round 4: this replaces the
// per-node segment sums (fp64 atomics per run and channel into seg[N][4]) AND node_stat_kernel, the O(N) kernel that turned
// them into the statistics: one launch less per round, 3 of the 25 of the S02 forward.  Any row order is correct; runs of
// length one (a randomly ordered list) pay one dependent 128-byte load per edge here -- the reference's lists are cartesian
// products, runs of hundreds.
template <int kEPT>
__global__ __launch_bounds__(256) void pass_b_kernel(RoundParams p) {
  drop_resolve(p.drop_e);
  __shared__ float s1[4], t1[4];
  __shared__ double red[14 * 4 + 4 * 128 + 4];
  const int lane = threadIdx.x & 63;
  // few-edge lists (one edge per thread) fold the statistics in as described; many-edge lists keep the per-node segment
  // sums + node_stat_kernel (kernels.h, fold_node_stat): pass B is issue-bound there and the fold costs it 15 instructions
  // per 64 edges
  constexpr bool kFold = kEPT == 1;
  const int k32 = lane & 31, hh = lane >> 5;             // C entries of this lane: channels (2 hh, 2 hh + 1) x column k32
  const float ub = p.un_b[k32];
  double c0 = 0, c1 = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * kEPT;
  const int64_t e_end = ((p.n_edges + 63) / 64) * 64;
  int64_t base = (int64_t)blockIdx.x * blockDim.x * kEPT + threadIdx.x;
  // the first trip's operands do not depend on the statistics: fetch them before waiting for those
  int rr[kEPT];
  float4 zz[kEPT];
  // a block's tile of 256 * kEPT edges is either whole (every tile but the list's last: no per-edge bounds checks, the
  // addresses are a block-uniform base + the thread index) or the last, partial one
  auto whole = [&](int64_t b) { return b - threadIdx.x + 256 * kEPT <= p.n_edges; };      // block-uniform
  auto fetch = [&](int64_t b) {
    if (whole(b)) {
      const int64_t t0 = b - threadIdx.x;
#pragma unroll
      for (int i = 0; i < kEPT; ++i) {
#if PB_NT
        typedef float f4v __attribute__((ext_vector_type(4)));
        rr[i] = __builtin_nontemporal_load(p.row32 + t0 + i * 256 + threadIdx.x);
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p.e_buf + (t0 + i * 256) * 4) + threadIdx.x);
        zz[i] = make_float4(v[0], v[1], v[2], v[3]);
#else
        rr[i] = (p.row32 + t0 + i * 256)[threadIdx.x];
        zz[i] = reinterpret_cast<const float4*>(p.e_buf + (t0 + i * 256) * 4)[threadIdx.x];
#endif
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      const int64_t e = b + i * 256;
      rr[i] = -1;
      zz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < p.n_edges) {
        rr[i] = p.row32[e];
        zz[i] = reinterpret_cast<const float4*>(p.e_buf)[e];
      }
    }
  };
  // qb[row][k32] of every wave-trip's first and last row (folded form): requested as soon as the row ids are there -- for the
  // first trip that is BEFORE the block waits for the z1 statistics, so the 128-byte lines travel during the prologue
  float qf[kEPT], ql[kEPT];
  auto prefetch_q = [&]() {
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      qf[i] = ql[i] = 0.f;
      if (kFold) {
        const int rf = __builtin_amdgcn_readfirstlane(rr[i]), rl = __builtin_amdgcn_readlane(rr[i], 63);
        qf[i] = p.Q[(int64_t)(rf < 0 ? 0 : rf) * kH + k32];
        ql[i] = p.Q[(int64_t)(rl < 0 ? 0 : rl) * kH + k32];
      }
    }
  };
  EK_T(3, 0);
  if (base < e_end) fetch(base);
  stat_gather(p.stats + kRoundZ1Off, 8, kZ1Stride, red);
  if (base < e_end) prefetch_q();
  __syncthreads();
  EK_T(3, 1);
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    bn_affine(red[k], red[4 + k], p.e_total, p.ue_g[k], p.ue_bt[k], s1[k], t1[k]);
  }
  __syncthreads();
  EK_T(3, 2);
  double acc[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i] = 0;
  bool first_trip = true;
  while (base < e_end) {
    const bool full = whole(base);
    float vv[kEPT][4];
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      const int64_t e = base + i * 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) vv[i][k] = 0.f;
      if (full || e < p.n_edges) {
        const float z4[4] = {zz[i].x, zz[i].y, zz[i].z, zz[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) vv[i][k] = fmaxf(fmaf(z4[k], s1[k], t1[k]), 0.f);
        drop_apply4(p.drop_e, p.drop_stream, (unsigned long long)e * 4, vv[i]);
        if (!p.lazy_e) reinterpret_cast<float4*>(p.e_out)[e] = make_float4(vv[i][0], vv[i][1], vv[i][2], vv[i][3]);
      }
    }
    // (later trips: requested before the moments are taken -- by the time the run sums exist the lines are there; rows of
    // inactive lanes: -1 -> row 0, never used)
    if (!first_trip) prefetch_q();
    first_trip = false;
    // (the readlane builtin is typed int: a float argument would be CONVERTED, not moved)
    auto rl_f32 = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
    // a run's channel sums (S0..S3, wave-uniform) times the row's qb: this lane's two entries of C
    auto add_run = [&](float S0, float S1, float S2, float S3, float qrow) {
      const double qb = (double)(qrow + ub);
      c0 = fma((double)(hh ? S2 : S0), qb, c0);
      c1 = fma((double)(hh ? S3 : S1), qb, c1);
    };
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      const int64_t e = base + i * 256;
      if (!full && e - lane >= p.n_edges) continue;   // whole wave past the end (wave-uniform)
      const bool active = full || e < p.n_edges;
      float* v = vv[i];
      const int r = rr[i];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a] += v[a];
#pragma unroll
        for (int b = a; b < 4; ++b) acc[4 + tri(4, a, b)] += (double)v[a] * v[b];
      }
      // Common case on row-sorted lists: the whole wave sits in one row -> one ten-instruction wave sum for the four
      // channels (lanes 15 / 31 / 47 / 63 end up with channels 0 / 2 / 1 / 3).
      const int r0 = __builtin_amdgcn_readfirstlane(r);
      const unsigned long long in_first = __ballot(r == r0);
      if (in_first == ~0ull && r0 >= 0) {
        const float t = wave_sum4_f32(v[0], v[1], v[2], v[3]);
        if (kFold) add_run(rl_f32(t, 15), rl_f32(t, 47), rl_f32(t, 31), rl_f32(t, 63), qf[i]);
        else if ((lane & 15) == 15) unsafeAtomicAdd(p.seg + (int64_t)r0 * 4 + wave_sum4_slot(lane), (double)t);
        continue;
      }
      // two rows in a whole wave (the next most common case on row-sorted lists at ~100 edges per row): the first row's
      // share and the total by the same sums, the second row's share as their difference (e' >= 0: no cancellation beyond
      // the total's rounding)
      const int r1 = __builtin_amdgcn_readlane(r, 63);
      if (r0 >= 0 && r1 >= 0 && (in_first | __ballot(r == r1)) == ~0ull) {
        const bool first = r == r0;
        const float tot = wave_sum4_f32(v[0], v[1], v[2], v[3]);
        const float fst = wave_sum4_f32(first ? v[0] : 0.f, first ? v[1] : 0.f, first ? v[2] : 0.f, first ? v[3] : 0.f);
        if (kFold) {
          const float f0 = rl_f32(fst, 15), f1 = rl_f32(fst, 47);
          const float f2 = rl_f32(fst, 31), f3 = rl_f32(fst, 63);
          add_run(f0, f1, f2, f3, qf[i]);
          add_run(rl_f32(tot, 15) - f0, rl_f32(tot, 47) - f1, rl_f32(tot, 31) - f2, rl_f32(tot, 63) - f3, ql[i]);
        } else if ((lane & 15) == 15) {
          const int k = wave_sum4_slot(lane);
          unsafeAtomicAdd(p.seg + (int64_t)r0 * 4 + k, (double)fst);
          unsafeAtomicAdd(p.seg + (int64_t)r1 * 4 + k, (double)(tot - fst));
        }
        continue;
      }
      // anything else (three or more rows, any order, a partial wave): segmented inclusive scan, then run by run
      const int prev = __shfl_up(r, 1, 64);
      int flag = (lane == 0 || prev != r) ? 1 : 0;
      const int next_head = __shfl_down(flag, 1, 64);
      const bool tail = active && (lane == 63 || next_head != 0 || e + 1 >= p.n_edges);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int f_up = __shfl_up(flag, off, 64);
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = __shfl_up(v[k], off, 64);
        if (lane >= off && !flag) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] += u[k];
          flag = f_up;
        }
      }
      if (!kFold) {
        if (tail) {
#pragma unroll
          for (int k = 0; k < 4; ++k) unsafeAtomicAdd(p.seg + (int64_t)r * 4 + k, (double)v[k]);
        }
        continue;
      }
      unsigned long long tails = __ballot(tail);          // a run's last lane holds the run's sums
      while (tails) {
        const int tl = __builtin_amdgcn_readfirstlane(__ffsll((long long)tails) - 1);
        tails &= tails - 1;
        const int rt = __builtin_amdgcn_readlane(r, tl);
        add_run(rl_f32(v[0], tl), rl_f32(v[1], tl), rl_f32(v[2], tl),
                rl_f32(v[3], tl), p.Q[(int64_t)rt * kH + k32]);
      }
    }
    base += stride;
    if (base < e_end) fetch(base);
  }
  EK_T(3, 3);
  // ---- the block's sums: moments of e' -> the M block; A_k . m1 and 2 A_k . C[:,k] -> the node-update (z2) block
  {
    const int wid = threadIdx.x >> 6;
    double* sm_m = red;                  // [4 waves][14]
    double* sm_c = red + 14 * 4;         // [4 waves][128]: C[j][k] at j * 32 + k
    double* sm_m1 = sm_c + 4 * 128;      // [4]
    wave_sums_f64<14>(acc, sm_m + wid * 14);            // (common.h: pairs folded by permlane swaps, 5 instructions per value)
    sm_c[wid * 128 + (2 * hh) * 32 + k32] = c0;
    sm_c[wid * 128 + (2 * hh + 1) * 32 + k32] = c1;
    __syncthreads();
    double* m_dst = p.stats + kRoundMOff + (blockIdx.x % kStatRep) * kMStride;
    if (threadIdx.x < 14) {
      const double t = sm_m[threadIdx.x] + sm_m[14 + threadIdx.x] + sm_m[28 + threadIdx.x] + sm_m[42 + threadIdx.x];
      unsafeAtomicAdd(m_dst + threadIdx.x, t);
      if (threadIdx.x < 4) sm_m1[threadIdx.x] = t;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 128) {   // (a wave of its own: not the one that writes sm_m1)
      const int i = threadIdx.x - 64;
      sm_c[i] = sm_c[i] + sm_c[128 + i] + sm_c[256 + i] + sm_c[384 + i];
    }
    __syncthreads();
    if (kFold && threadIdx.x < 32) {
      const int k = threadIdx.x;
      const float* a = p.un_w + k * p.un_ld + p.un_eoff;
      double lin = 0, cross = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lin += (double)a[j] * sm_m1[j];
        cross += (double)a[j] * sm_c[j * 32 + k];
      }
      double* z_dst = p.stats + kRoundZ2Off + (blockIdx.x % kStatRep) * kZ2Stride;
      unsafeAtomicAdd(z_dst + k, lin);
      unsafeAtomicAdd(z_dst + 32 + k, 2.0 * cross);
    }
  }
  EK_T(3, 4);
}

// ------------------------------------------------------------------------------------------------
// pass C: 32 lanes = 32 channels; every half-wave walks 32 consecutive edges of the block's tile
// ------------------------------------------------------------------------------------------------

// Order-independent aggregation for row-sorted lists (MTMC_F_DETERMINISTIC): a 32-edge chunk's first run goes to
// carry[chunk][0], its last run (if different) to carry[chunk][1], runs strictly inside the chunk are complete rows
// and are stored directly; agg_fixup_kernel then adds each row's pieces in chunk order.  slot: 0 / 1 / -1 (inside).
__device__ __forceinline__ void flush_run(const RoundParams& p, bool det, int64_t chunk, int slot, int node, int k,
                                          float acc);

__device__ __forceinline__ void flush_node(const RoundParams& p, int node, int k, float acc) {
  float* dst = p.h_acc + (int64_t)node * kH + k;
  if (p.agg == 2) {
    atomicMax(reinterpret_cast<int*>(dst), __float_as_int(acc));   // values are >= 0: int order == float order
  } else {
    unsafeAtomicAdd(dst, acc);
  }
}

__device__ __forceinline__ void flush_run(const RoundParams& p, bool det, int64_t chunk, int slot, int node, int k,
                                          float acc) {
  if (!det) { flush_node(p, node, k, acc); return; }
  if (slot < 0) p.h_acc[(int64_t)node * kH + k] = acc;
  else p.carry[(chunk * 2 + slot) * kH + k] = acc;
}

__global__ __launch_bounds__(256) void agg_fixup_kernel(RoundParams p) {
  if (p.flags[0] != 0) return;                       // unsorted rows: pass C used atomics, nothing to add up
  const int k = threadIdx.x & 31;
  const int64_t stride = (int64_t)gridDim.x * 8;
  for (int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); i < p.n_nodes; i += stride) {
    const int d = p.deg[i];
    if (d == 0) continue;
    const int64_t s = p.row_start[i], t = s + d, len = p.det_len;      // len: edges per carry chunk (walk: 32; sorted: a span)
    const int64_t cs = s / len, ce = (t - 1) / len;
    float v;
    if (cs == ce) {
      if (s % len == 0) v = p.carry[(cs * 2 + 0) * kH + k];
      else if (t % len == 0 || t == p.n_edges) v = p.carry[(cs * 2 + 1) * kH + k];
      else continue;                                 // strictly inside its chunk: already stored
    } else {
      v = p.carry[(cs * 2 + (s % len == 0 ? 0 : 1)) * kH + k];
      for (int64_t c = cs + 1; c <= ce; ++c) v += p.carry[(c * 2 + 0) * kH + k];
    }
    p.h_acc[i * kH + k] = v;
  }
}

typedef float f32x16c __attribute__((ext_vector_type(16)));
constexpr int kTileC = 256;   // tile = block size (128 and 512 measured slower at every graph size)
__global__ __launch_bounds__(kTileC) void pass_c_kernel(RoundParams p) {
  drop_resolve(p.drop_n);
  __shared__ float4 tile_e[kTileC];
  __shared__ int tile_row[kTileC];
  __shared__ double st[10 + 64];           // e' second moments (10) | z2 sums (32) | z2 sums of squares (32)
  const int k = threadIdx.x & 31;          // channel
  const int hw = threadIdx.x >> 5;         // half-wave
  // many-edge row-sorted list: pass_c_sorted_kernel, launched just before, did every whole 64-edge chunk; what is left for
  // this kernel are the < 64 edges behind the last one
  int64_t e_first = 0;
  if ((p.mfma_c == 1 || p.mfma_c == 4) && p.flags[0] == 0) return;  // the matrix-core kernel did all of a sorted list
                                                                     // (1: MTMC_PASS_C_GENERAL, 4: deterministic mode)
  if (p.mfma_c == 3 && p.flags[0] == 0) {
    e_first = p.n_edges & ~(int64_t)63;
    if (e_first == p.n_edges) return;
  }
  const int64_t n_tiles = (p.n_edges + kTileC - 1) / kTileC;
  int64_t tile = e_first / kTileC + blockIdx.x;
  if (tile >= n_tiles) return;
  // a tile's operands do not depend on the statistics: fetch the first one before waiting for those
  float4 ev;
  int rw;
  auto fetch = [&](int64_t t) {
    const int64_t e = t * kTileC + threadIdx.x;
    ev = make_float4(0, 0, 0, 0);
    rw = 0;
    if (e < p.n_edges) {
      ev = reinterpret_cast<const float4*>(p.e_out)[e];
      rw = p.row32[e];
    }
  };
  if (tile < n_tiles) fetch(tile);
  // BatchNorm affine of channel k of z2 from the moment statistics (node_stat_kernel + pass B)
  stat_gather2(p.stats + kRoundMOff + 4, 10, kMStride, p.stats + kRoundZ2Off, 64, kZ2Stride, st);
  __shared__ double st1[8];
  __shared__ float s1[4], t1[4];
  if (p.lazy_e) stat_gather(p.stats + kRoundZ1Off, 8, kZ1Stride, st1);
  __syncthreads();
  if (p.lazy_e) {                          // the edge buffer holds z1: e' = relu(s1 z1 + t1), recomputed here
    if (threadIdx.x < 4)
      bn_affine(st1[threadIdx.x], st1[4 + threadIdx.x], p.e_total, p.ue_g[threadIdx.x], p.ue_bt[threadIdx.x],
                s1[threadIdx.x], t1[threadIdx.x]);
    __syncthreads();
  }
  float sk, tk;
  {
    const float* a = p.un_w + k * p.un_ld + p.un_eoff;
    const double quad = quad_form(a, 4, st);
    bn_affine(st[10 + k], st[10 + 32 + k] + quad, p.e_total, p.un_g[k], p.un_bt[k], sk, tk);
  }
  float a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = sk * p.un_w[k * p.un_ld + p.un_eoff + j];
  const float cb = fmaf(sk, p.un_b[k], tk);
  const bool det = p.det && p.agg != 2 && p.flags[0] == 0;      // max is order-independent as it is

  while (tile < n_tiles) {
    const int64_t base = tile * kTileC;
    const int64_t e = base + threadIdx.x;
    if (p.lazy_e) {
      ev.x = fmaxf(fmaf(ev.x, s1[0], t1[0]), 0.f); ev.y = fmaxf(fmaf(ev.y, s1[1], t1[1]), 0.f);
      ev.z = fmaxf(fmaf(ev.z, s1[2], t1[2]), 0.f); ev.w = fmaxf(fmaf(ev.w, s1[3], t1[3]), 0.f);
    }
    if (e < p.n_edges) {
      tile_row[threadIdx.x] = rw;
      if (p.logits && e >= e_first) {                         // classifier on this edge (mpn.py:291-292)
        float lg[MTMC_MAX_CLASSES];
        for (int c = 0; c < p.n_classes; ++c) {
          const float* w = p.cls_w + c * 4;
          lg[c] = fmaf(w[3], ev.w, fmaf(w[2], ev.z, fmaf(w[1], ev.y, fmaf(w[0], ev.x, p.cls_b[c]))));
        }
        if (p.n_classes == 2) {
          reinterpret_cast<float2*>(p.logits)[e] = make_float2(lg[0], lg[1]);
        } else {
          for (int c = 0; c < p.n_classes; ++c) p.logits[e * p.n_classes + c] = lg[c];
        }
      }
    }
    tile_e[threadIdx.x] = ev;
    __syncthreads();
    const int n_here = base + hw * 32 < e_first ? 0 : (int)min((int64_t)32, p.n_edges - (base + hw * 32));   // may be <= 0
    const int64_t chunk = (base >> 5) + hw;
    int cur = -1;
    bool first = true;
    float acc = 0.f, c0 = 0.f;
    unsigned long long zd = 0;               // Dropout: the hash of four consecutive edges of this lane's channel (common.h)
    for (int j = 0; j < n_here; ++j) {
      const int r = tile_row[hw * 32 + j];
      const float4 v = tile_e[hw * 32 + j];
      if (p.drop_n.on && (j & 3) == 0)
        zd = drop_hash4(p.drop_n.seed, p.drop_stream + 1, drop_msg_index(base + hw * 32 + j, k, p.n_edges) >> 2);
      if (r != cur) {
        if (cur >= 0) { flush_run(p, det, chunk, first ? 0 : -1, cur, k, acc); first = false; }
        cur = r;
        acc = 0.f;
        c0 = fmaf(sk, p.Q[(int64_t)r * kH + k], cb);
      }
      float m = fmaxf(fmaf(a4[3], v.w, fmaf(a4[2], v.z, fmaf(a4[1], v.y, fmaf(a4[0], v.x, c0)))), 0.f);
      if (p.drop_n.on) m = drop_field(p.drop_n, zd, j & 3) ? m * p.drop_n.inv_keep : 0.f;
      acc = (p.agg == 2) ? fmaxf(acc, m) : acc + m;
    }
    if (cur >= 0) flush_run(p, det, chunk, first ? 0 : 1, cur, k, acc);
    tile += gridDim.x;
    if (tile < n_tiles) fetch(tile);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// pass C for many-edge, row-sorted lists: the 32-wide node-update pre-activation of 32 edges at a time on the matrix
// cores.  A resident grid of waves takes spans of span_c consecutive 64-edge chunks round-robin.  Per chunk: one coalesced float4 (e') + row id per lane; two
// v_permlane32_swap turn the four components into the A operands of v_mfma_f32_32x32x2_f32 (A[i][k] = e'_k of edge i),
// the B operand is the channel's BatchNorm-scaled weight row (constant per lane), so
//     Z[32 edges][32 channels] = E' . (s_k A_k)^T             2 MFMAs per 32 edges (K = 4)
// arrives as 16 registers per lane = 16 edges of the lane's channel.  The ReLU and the per-row sums are then 3 VALU ops
// per register (c0 = s_k (Q[row][k] + b_k) + t_k is one value per lane while the group stays in one row -- the common
// case at the average degrees this kernel is launched for); runs are carried in registers across groups and chunks and
// flushed with one 128-byte row of float atomics when the row changes.  Groups that straddle rows take a masked pass
// per distinct row.  Exact same arithmetic per edge as pass_c_kernel (fp32 FMAs in k order inside the MFMA); sum / mean
// aggregation, eval mode, non-deterministic mode only -- everything else stays on pass_c_kernel.
// The walk in pass_c_kernel costs ~4.5 wave-instructions per edge; this one ~1.5.
// ------------------------------------------------------------------------------------------------
// 64-edge chunks per wave span (runs are carried inside a span; one span per wave, so the grid balances itself)

template <bool DROP>        // DROP (training, few-edge lists): the node update's Dropout on the messages, per accumulator register
__global__ __launch_bounds__(256, (DROP ? 3 : 1)) void pass_c_mfma_kernel(RoundParams p, int span_c) {   // DROP: 168 registers, three blocks per CU
  if (DROP) drop_resolve(p.drop_n);
  __shared__ double st[10 + 64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: spans, chunks, rows stay scalar
  const int k = lane & 31, hi = lane >> 5;
  if (p.mfma_c == 1 && p.flags[0] != 0) return;              // unsorted rows: pass_c_kernel does this round (block-uniform)
  EK_T(4, 0);
  stat_gather2(p.stats + kRoundMOff + 4, 10, kMStride, p.stats + kRoundZ2Off, 64, kZ2Stride, st);
  __shared__ double st1[8];
  __shared__ float s1s[4], t1s[4];
  if (p.lazy_e) stat_gather(p.stats + kRoundZ1Off, 8, kZ1Stride, st1);
  __syncthreads();
  EK_T(4, 1);
  float s1[4] = {1.f, 1.f, 1.f, 1.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.lazy_e) {                                            // the edge buffer holds z1: e' = relu(s1 z1 + t1), recomputed here
    if (threadIdx.x < 4)
      bn_affine(st1[threadIdx.x], st1[4 + threadIdx.x], p.e_total, p.ue_g[threadIdx.x], p.ue_bt[threadIdx.x],
                s1s[threadIdx.x], t1s[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = s1s[j]; t1[j] = t1s[j]; }
  }
  float sk, tk;
  {
    const float* a = p.un_w + k * p.un_ld + p.un_eoff;
    const double quad = quad_form(a, 4, st);
    bn_affine(st[10 + k], st[10 + 32 + k] + quad, p.e_total, p.un_g[k], p.un_bt[k], sk, tk);
  }
  const float b0 = sk * p.un_w[k * p.un_ld + p.un_eoff + hi];          // B[kk = hi][j = k] of the two K = 2 steps
  const float b1 = sk * p.un_w[k * p.un_ld + p.un_eoff + 2 + hi];
  const float cb = fmaf(sk, p.un_b[k], tk);
  const bool want_logits = p.logits != nullptr;
  float cw[2][4] = {}, cbias[2] = {};
  if (want_logits && p.n_classes == 2)
    for (int c = 0; c < 2; ++c) {
      cbias[c] = p.cls_b[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) cw[c][j] = p.cls_w[c * 4 + j];
    }

  EK_T(4, 2);
  const int64_t n_chunks = (p.n_edges + 63) / 64;
  const int64_t n_spans = (n_chunks + span_c - 1) / span_c;
  const int64_t wave_stride = (int64_t)gridDim.x * 4;
  for (int64_t span = (int64_t)blockIdx.x * 4 + wid; span < n_spans; span += wave_stride) {
    int cur = -1;                 // wave-uniform: the row whose partial sum `run` holds (this lane's channel, this half's edges)
    float run = 0.f;
    auto flush = [&]() {
      if (cur >= 0) {
        const float tot = run + __shfl_xor(run, 32, 64);
        if (hi == 0) unsafeAtomicAdd(p.h_acc + (int64_t)cur * kH + k, tot);
      }
    };
    const int64_t c_end = min(n_chunks, (span + 1) * span_c);
    // A three-deep software pipeline of PLAIN loads (the compiler counts every one of them, so its own s_waitcnt is exact;
    // round 2 issued the Q lookups from inline asm behind a hand-counted s_waitcnt vmcnt(2)):
    //   chunk c + 2: e' / row ids requested                                  (ev_3, rw_3)
    //   chunk c + 1: row ids have landed -> the Q rows of its first and last row requested   (all of its rows at the degrees
    //                this kernel runs at; other rows of a chunk are looked up on demand)
    //   chunk c    : everything is in registers -> classifier, MFMAs, sums
    // so a k-iteration waits once, at its top, for loads issued a whole iteration earlier, and the table lookups' latency
    // (the Q table lives in L2 / Infinity Cache: 0.3-1 us under load) no longer sits between a chunk's loads and its MFMAs.
    auto fetch = [&](int64_t chunk, float4& ev_o, int& rw_o) {    // clamped address: chunks past the end re-read the last edge
      const int64_t e = min(chunk * 64 + lane, p.n_edges - 1);
      ev_o = reinterpret_cast<const float4*>(p.e_out)[e];
      rw_o = p.row32[e];
    };
    auto rows_of = [&](int rw_raw, int64_t chunk, int& first, int& last) {   // first / last row of a chunk (wave-uniform)
      const int nv = (int)min((int64_t)64, p.n_edges - chunk * 64);
      first = __builtin_amdgcn_readfirstlane(rw_raw);
      last = __builtin_amdgcn_readfirstlane(__shfl(rw_raw, (nv > 0 ? nv : 1) - 1, 64));
    };
    int qa_row = -1, qb_row = -1;
    float qa = 0.f, qb = 0.f;
    auto c0_of = [&](int r) -> float {                         // r wave-uniform
      const float q = r == qa_row ? qa : (r == qb_row ? qb : p.Q[(int64_t)r * kH + k]);
      return fmaf(sk, q, cb);
    };
    auto account = [&](int r, float part) {                     // add a group's partial sum to row r's run
      if (r != cur) {
        flush();
        cur = r;
        run = 0.f;
      }
      run += part;
    };
    // sum of relu over the lane's 16 values.  relu as a signed-integer max: exact for every float (negative values and
    // -0 have the sign bit set), and free of the canonicalising v_max_f32 x, x, x that fmaxf puts in front
    // DROP: keep[g][i] = 1/(1-p) or 0 for register i of group g (edge 64 chunk + 32 g + (i&3) + 8 (i>>2) + 4 hi, channel k:
    // drop_msg_index, what the walk and the backward hash too), taken once per chunk: eight hashes for the 32 registers
    unsigned keepbits[2] = {0u, 0u};                 // (bit i: register i kept -- one register instead of 32 factors)
    const float ikn = DROP ? p.drop_n.inv_keep : 1.f;
    auto keepf = [&](int g, int i) -> float { return ((keepbits[g] >> i) & 1u) ? ikn : 0.f; };
    auto relu_sum = [&](const f32x16c& acc, int g) -> float {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float t0 = __int_as_float(max(__float_as_int(acc[i]), 0)), t1 = __int_as_float(max(__float_as_int(acc[i + 1]), 0));
        s0 += DROP ? t0 * keepf(g, i) : t0;
        s1 += DROP ? t1 * keepf(g, i + 1) : t1;
      }
      return s0 + s1;
    };
    const int64_t c0_chunk = span * span_c;
    float4 ev_1, ev_2;
    int rw_1, rw_2;
    // (requesting the wave's first chunk BEFORE the statistics are gathered -- it does not depend on them -- was measured in
    //  round 5 and took 1.6 us MORE per launch on the 150k-edge graph, profiles/r05_s02_edge_trims.txt: not kept)
    fetch(c0_chunk, ev_1, rw_1);
    rows_of(rw_1, c0_chunk, qa_row, qb_row);
    qa = p.Q[(int64_t)qa_row * kH + k];
    qb = p.Q[(int64_t)qb_row * kH + k];
    // Nothing is requested for chunks past the span's end: on few-edge lists a span is ONE chunk, and waiting for the row ids
    // of a next chunk that does not exist was a whole round trip in front of the only chunk's MFMAs (round 5 stamps,
    // profiles/r05_edge_stamps.txt: 7.0 -> 5.7 us per launch)
    ev_2 = ev_1; rw_2 = rw_1;
    if (c0_chunk + 1 < c_end) fetch(c0_chunk + 1, ev_2, rw_2);
    for (int64_t chunk = c0_chunk; chunk < c_end; ++chunk) {
      // ---- chunk + 1: its row ids were requested one iteration ago; request its Q rows.  chunk + 2: request the stream.
      int qa_row_2 = qa_row, qb_row_2 = qb_row;
      float qa_2 = qa, qb_2 = qb;
      if (chunk + 1 < c_end) {
        rows_of(rw_2, chunk + 1, qa_row_2, qb_row_2);
        qa_2 = p.Q[(int64_t)qa_row_2 * kH + k];
        qb_2 = p.Q[(int64_t)qb_row_2 * kH + k];
      }
      float4 ev_3 = ev_2;
      int rw_3 = rw_2;
      if (chunk + 2 < c_end) fetch(chunk + 2, ev_3, rw_3);
      // ---- chunk: all in registers
      const int64_t e = chunk * 64 + lane;
      const bool valid = e < p.n_edges;
      float4 ev = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        ev = ev_1;
        if (p.lazy_e) {
          ev.x = fmaxf(fmaf(ev.x, s1[0], t1[0]), 0.f); ev.y = fmaxf(fmaf(ev.y, s1[1], t1[1]), 0.f);
          ev.z = fmaxf(fmaf(ev.z, s1[2], t1[2]), 0.f); ev.w = fmaxf(fmaf(ev.w, s1[3], t1[3]), 0.f);
        }
      }
      const int rw = valid ? rw_1 : -1;
      const int n_valid = (int)min((int64_t)64, p.n_edges - chunk * 64);      // scalar
      if (DROP) {
        keepbits[0] = keepbits[1] = 0u;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) {                       // registers 4q .. 4q+3: four consecutive edges, one hash
            const unsigned long long zd =
                drop_hash4(p.drop_n.seed, p.drop_stream + 1, drop_msg_index(chunk * 64 + 32 * g + 8 * q + 4 * hi, k, p.n_edges) >> 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) keepbits[g] |= drop_field(p.drop_n, zd, j) ? (1u << (4 * q + j)) : 0u;
          }
      }
      float lg0 = 0.f, lg1 = 0.f;
      if (want_logits && p.n_classes == 2) {                    // classifier on this edge (mpn.py:291-292): arithmetic now,
        // two scalar FMA chains, kept apart: paired up by the SLP vectoriser they become v_pk_fma_f32 with op_sel,
        // the form tools/check_isa.py bans from kernels with MFMAs
        lg0 = fmaf(cw[0][3], ev.w, fmaf(cw[0][2], ev.z, fmaf(cw[0][1], ev.y, fmaf(cw[0][0], ev.x, cbias[0]))));
        asm volatile("" : "+v"(lg0));
        lg1 = fmaf(cw[1][3], ev.w, fmaf(cw[1][2], ev.z, fmaf(cw[1][1], ev.y, fmaf(cw[1][0], ev.x, cbias[1]))));
      }
      // A operands: lanes 0..31 carry component k = 0 (2), lanes 32..63 component k = 1 (3) of edge (lane & 31)
      const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(ev.x), __float_as_uint(ev.y), false, false);
      const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ev.z), __float_as_uint(ev.w), false, false);
      const unsigned long long row_first_mask = __ballot(rw == qa_row), row_last_mask = __ballot(rw == qb_row);
      // group g = edges 32g .. 32g+31 of the chunk.  At the degrees this kernel runs at a group touches one row, or two
      // (a row boundary inside it): classify each group by its first row (mask ma), the row of its first other edge
      // (mask mb), and whether those two cover it.  All of this is scalar.
      const int nv1 = n_valid - 32;
      const unsigned gmv[2] = {n_valid >= 32 ? 0xffffffffu : ((1u << n_valid) - 1u),
                               nv1 >= 32 ? 0xffffffffu : (nv1 > 0 ? ((1u << nv1) - 1u) : 0u)};
      int ra[2], rb[2];
      unsigned ma[2], mb[2];
      bool two_rows_max[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        ra[g] = rb[g] = -1;
        ma[g] = mb[g] = 0;
        two_rows_max[g] = true;
        if (gmv[g] == 0) continue;
        ra[g] = g == 0 ? qa_row : __builtin_amdgcn_readlane(rw, 32);
        const unsigned long long fa = g == 0 ? row_first_mask : (ra[g] == qb_row ? row_last_mask : __ballot(rw == ra[g]));
        ma[g] = (unsigned)(fa >> (32 * g)) & gmv[g];
        const unsigned rest = gmv[g] & ~ma[g];
        if (rest != 0) {
          rb[g] = __builtin_amdgcn_readfirstlane(__shfl(rw, 32 * g + __ffs(rest) - 1, 64));
          const unsigned long long fb = rb[g] == qb_row ? row_last_mask : __ballot(rw == rb[g]);
          mb[g] = (unsigned)(fb >> (32 * g)) & gmv[g];
          two_rows_max[g] = (ma[g] | mb[g]) == gmv[g];
        }
      }
      if (want_logits && valid) {
        if (p.n_classes == 2) {
          reinterpret_cast<float2*>(p.logits)[e] = make_float2(lg0, lg1);
        } else {
          for (int c = 0; c < p.n_classes; ++c) {
            const float* w = p.cls_w + c * 4;
            p.logits[e * p.n_classes + c] = fmaf(w[3], ev.w, fmaf(w[2], ev.z, fmaf(w[1], ev.y, fmaf(w[0], ev.x, p.cls_b[c]))));
          }
        }
      }
      if (two_rows_max[0] && two_rows_max[1]) {
        // Z = c0[row of edge] + E' . (s A)^T: the row constants ride in as a third K step whose A operand is the
        // membership of edge i in the group's first / second row and whose B operand holds the two constants; both
        // groups' chains are issued before either result is consumed
        f32x16c acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned long long member = (unsigned long long)ma[g] | ((unsigned long long)mb[g] << 32);
          const float a_c = __builtin_amdgcn_inverse_ballot_w64(member) ? 1.0f : 0.0f;
          const float c_a = ra[g] >= 0 ? c0_of(ra[g]) : 0.f, c_b = rb[g] >= 0 ? c0_of(rb[g]) : 0.f;
          f32x16c z = {};
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, hi == 0 ? c_a : c_b, z, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xy[g]), b0, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zw[g]), b1, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (ra[g] < 0) break;
          const float total = relu_sum(acc[g], g);
          if (rb[g] < 0) {
            account(ra[g], total);
          } else {                                             // the first row's share through per-register lane masks
            float sa0 = 0.f, sa1 = 0.f;                        // (register i = edge (i&3) + 8(i>>2) + 4*hi of the group)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int c = (i & 3) + 8 * (i >> 2);
              const unsigned long long in_a = (((ma[g] >> c) & 1u) ? 0xffffffffull : 0ull) |
                                              (((ma[g] >> (c + 4)) & 1u) ? 0xffffffff00000000ull : 0ull);
              const float t0 = __int_as_float(max(__float_as_int(acc[g][i]), 0));
              const float t = DROP ? t0 * keepf(g, i) : t0;
              const float v = __builtin_amdgcn_inverse_ballot_w64(in_a) ? t : 0.f;
              if (i & 1) sa1 += v; else sa0 += v;
            }
            const float sa = sa0 + sa1;
            account(ra[g], sa);
            account(rb[g], total - sa);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned gm = gmv[g];
          if (gm == 0) break;
          f32x16c acc = {};
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xy[g]), b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zw[g]), b1, acc, 0, 0, 0);
          // how many distinct rows?  Up to four take one masked pass each; a group that is all over the place (unsorted
          // or low-degree stretch of the list) is cheaper per EDGE: every register adds its own edge's constant and goes
          // to its row with one atomic (bounded cost, no run bookkeeping)
          int n_rows = 0;
          {
            unsigned left = gm;
            while (left != 0 && n_rows <= 4) {
              const int r = __builtin_amdgcn_readfirstlane(__shfl(rw, 32 * g + __ffs(left) - 1, 64));
              left &= ~((unsigned)(__ballot(rw == r) >> (32 * g)));
              ++n_rows;
            }
          }
          if (n_rows > 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int off = (i & 3) + 8 * (i >> 2) + 4 * hi;
              const int r = __shfl(rw, 32 * g + off, 64);
              if (r >= 0 && ((gm >> off) & 1u)) {
                const float c0 = fmaf(sk, p.Q[(int64_t)r * kH + k], cb);
                unsafeAtomicAdd(p.h_acc + (int64_t)r * kH + k, DROP ? fmaxf(acc[i] + c0, 0.f) * keepf(g, i) : fmaxf(acc[i] + c0, 0.f));
              }
            }
            continue;
          }
          unsigned done = 0;
          while (done != gm) {                                   // one masked pass per distinct row of the group
            const int pos = __ffs(gm & ~done) - 1;
            const int r = __builtin_amdgcn_readfirstlane(__shfl(rw, 32 * g + pos, 64));
            const unsigned same = (unsigned)(__ballot(rw == r) >> (32 * g)) & gm & ~done;
            done |= same;
            const float c0 = c0_of(r);
            const unsigned mine = same >> (4 * hi);             // bit (i&3) + 8(i>>2): register i's edge of this half-wave
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float t = DROP ? fmaxf(acc[i] + c0, 0.f) * keepf(g, i) : fmaxf(acc[i] + c0, 0.f);
              sacc += ((mine >> ((i & 3) + 8 * (i >> 2))) & 1u) ? t : 0.f;
            }
            account(r, sacc);
          }
        }
      }
      // rotate the pipeline
      ev_1 = ev_2; rw_1 = rw_2;
      ev_2 = ev_3; rw_2 = rw_3;
      qa_row = qa_row_2; qb_row = qb_row_2;
      qa = qa_2; qb = qb_2;
    }
    flush();
  }
  EK_T(4, 3);
}

// ------------------------------------------------------------------------------------------------
// pass C for MANY-edge ROW-SORTED lists (plan 1): the kernel above with everything a sorted list of whole 64-edge chunks
// makes unnecessary taken out of the loop -- counters of round 3 (r03: SQ_INSTS_*) showed that kernel bound by instruction
// issue, 189 vector + 142 scalar instructions per chunk, most of them bookkeeping:
//  * only WHOLE chunks here (the < 64 edges behind the last whole chunk go to pass_c_kernel, launched behind this one):
//    no clamped 64-bit addresses, no validity masks, a chunk's first / last rows are v_readlane of fixed lanes;
//  * sorted rows: a group's first row is its lane 0, its last row its lane 31, and the group is covered by those two iff
//    ballot(row == first) | ballot(row == last) is everything -- no search for "the first other edge";
//  * the row constants c0 = s_k (Q[row][k] + b_k) + t_k ride in as the B operand of a K step whose A operand is the
//    membership in the group's first (kk = 0) / last (kk = 1) row: lanes 0-31 of a B operand hold kk = 0, lanes 32-63 kk = 1,
//    so ONE Q lookup per group -- row = lane < 32 ? first : last -- is that operand; no per-row lookups, no selects;
//  * a two-row group's first share: its edges are a PREFIX of the group, register i of a lane holds edge (i&3) + 8(i>>2) +
//    4 hi, so whole 4-register blocks lie on one side and only block n_first >> 3 is cut: block sums (needed for the total
//    anyway) + four compares in the one cut block, instead of sixteen scalar-built lane masks;
//  * the three-deep pipeline is unrolled by three with the stage registers renamed: no register moves.
// Groups that touch three or more rows (low-degree stretches) take the masked pass per distinct row, as above.
// ------------------------------------------------------------------------------------------------
// DET (MTMC_F_DETERMINISTIC): no atomics.  A span's first run goes to carry[span][0], its last run (if different) to
// carry[span][1], runs strictly inside a span are complete rows and are stored; agg_fixup_kernel adds a row's pieces in span
// order (the walk's scheme, flush_run above, with spans of span_c * 64 edges for its 32-edge chunks).  The < 64 edges behind
// the last whole chunk are then taken here too, by the wave that owns the last span, through the masked passes.
template <bool LAZY, bool DET>
__global__ __launch_bounds__(256) void pass_c_sorted_kernel(RoundParams p, int span_c) {
  __shared__ double st[10 + 64];
  __shared__ double st1[8];
  __shared__ float s1s[4], t1s[4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = lane & 31, hi = lane >> 5;
  if (p.flags[0] != 0) return;                                // unsorted rows: pass_c_kernel does this round (block-uniform)
  stat_gather2(p.stats + kRoundMOff + 4, 10, kMStride, p.stats + kRoundZ2Off, 64, kZ2Stride, st);
  if (LAZY) stat_gather(p.stats + kRoundZ1Off, 8, kZ1Stride, st1);
  __syncthreads();
  float s1[4] = {1.f, 1.f, 1.f, 1.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
  if (LAZY) {                                                 // the edge buffer holds z1: e' = relu(s1 z1 + t1), recomputed here
    if (threadIdx.x < 4)
      bn_affine(st1[threadIdx.x], st1[4 + threadIdx.x], p.e_total, p.ue_g[threadIdx.x], p.ue_bt[threadIdx.x],
                s1s[threadIdx.x], t1s[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = s1s[j]; t1[j] = t1s[j]; }
  }
  float sk, tk;
  {
    const float* a = p.un_w + k * p.un_ld + p.un_eoff;
    const double quad = quad_form(a, 4, st);
    bn_affine(st[10 + k], st[10 + 32 + k] + quad, p.e_total, p.un_g[k], p.un_bt[k], sk, tk);
  }
  const float b0 = sk * p.un_w[k * p.un_ld + p.un_eoff + hi];          // B[kk = hi][j = k] of the two K = 2 steps
  const float b1 = sk * p.un_w[k * p.un_ld + p.un_eoff + 2 + hi];
  const float cb = fmaf(sk, p.un_b[k], tk);
  const bool two_class = p.logits != nullptr && p.n_classes == 2;
  float cw[2][4] = {}, cbias[2] = {};
  if (two_class)
    for (int c = 0; c < 2; ++c) {
      cbias[c] = p.cls_b[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) cw[c][j] = p.cls_w[c * 4 + j];
    }
  const int lim_hi = 4 * hi;
  const unsigned koff = (unsigned)k * 4u;

  const int n_full = (int)(p.n_edges >> 6);                  // whole chunks (launch_pass_c: fewer than 2^31 of them)
  const int n_tail = (int)(p.n_edges & 63);                  // edges behind them (DET: taken here; else by pass_c_kernel)
  const int n_spans = ((DET && n_tail ? n_full + 1 : n_full) + span_c - 1) / span_c;
  const int wave_stride = (int)gridDim.x * 4;
  for (int span = (int)blockIdx.x * 4 + wid; span < n_spans; span += wave_stride) {
    int cur = -1;                 // wave-uniform: the row whose partial sum `run` holds (this lane's channel, this half's edges)
    float run = 0.f;
    bool first_run = true;        // DET: nothing of this span has been flushed yet
    auto flush = [&](bool last) {
      if (cur >= 0) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(run), __float_as_uint(run), false, false);
        const float tot = run + __uint_as_float(hi == 0 ? sw[1] : sw[0]);
        if (!DET) {
          if (hi == 0) unsafeAtomicAdd(p.h_acc + (int64_t)cur * kH + k, tot);
        } else if (hi == 0) {
          if (first_run) p.carry[((int64_t)span * 2 + 0) * kH + k] = tot;
          else if (last) p.carry[((int64_t)span * 2 + 1) * kH + k] = tot;
          else p.h_acc[(int64_t)cur * kH + k] = tot;             // a row that began and ended inside this span
        }
        first_run = false;
      }
    };
    auto account = [&](int r, float part) {                     // add a group's partial sum to row r's run
      if (r != cur) {
        flush(false);
        cur = r;
        run = 0.f;
      }
      run += part;
    };
    // groups that are not one or two rows: a masked pass per distinct row (up to four; DET: any number), else every
    // register goes to its own row with one atomic (bounded cost, no run bookkeeping).  gm: the group's valid edges.
    auto slow_group = [&](const f32x16c& acc, int rwc, int g, unsigned gm) {
      int n_rows = 0;
      if (!DET) {
        unsigned left = gm;
        while (left != 0 && n_rows <= 4) {
          const int r = __builtin_amdgcn_readfirstlane(__shfl(rwc, 32 * g + __ffs(left) - 1, 64));
          left &= ~((unsigned)(__ballot(rwc == r) >> (32 * g)));
          ++n_rows;
        }
      }
      if (!DET && n_rows > 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int off = (i & 3) + 8 * (i >> 2) + 4 * hi;
          const int r = __shfl(rwc, 32 * g + off, 64);
          if ((gm >> off) & 1u) {
            const float c0 = fmaf(sk, p.Q[(int64_t)r * kH + k], cb);
            unsafeAtomicAdd(p.h_acc + (int64_t)r * kH + k, fmaxf(acc[i] + c0, 0.f));
          }
        }
        return;
      }
      unsigned done = 0;
      while (done != gm) {
        const int pos = __ffs(gm & ~done) - 1;
        const int r = __builtin_amdgcn_readfirstlane(__shfl(rwc, 32 * g + pos, 64));
        const unsigned same = (unsigned)(__ballot(rwc == r) >> (32 * g)) & gm & ~done;
        done |= same;
        const float c0 = fmaf(sk, p.Q[(int64_t)r * kH + k], cb);
        const unsigned mine = same >> (4 * hi);               // bit (i&3) + 8(i>>2): register i's edge of this half-wave
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float tt = fmaxf(acc[i] + c0, 0.f);
          sacc += ((mine >> ((i & 3) + 8 * (i >> 2))) & 1u) ? tt : 0.f;
        }
        account(r, sacc);
      }
    };
    const int c_beg = span * span_c, c_end = min(n_full, c_beg + span_c);
    // stage registers: chunk c lives in slot c % 3 (literal at every use after inlining)
    float4 ev[3];
    int rw[3];
    float qv[3][2];              // per group: Q[lane < 32 ? first row : last row][k]
    int rows[3][4];              // first / last row of group 0, of group 1 (scalar)
    auto fetch = [&](int c, int slot) {                         // whole chunks only: c is clamped to the last one
      const int cc = min(c, n_full - 1);
      const int64_t e0 = (int64_t)cc * 64;
#if PC_NT
      {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p.e_out + e0 * 4) + lane);
        ev[slot] = make_float4(v[0], v[1], v[2], v[3]);
        rw[slot] = __builtin_nontemporal_load(p.row32 + e0 + lane);
      }
#else
      ev[slot] = reinterpret_cast<const float4*>(p.e_out + e0 * 4)[lane];
      rw[slot] = (p.row32 + e0)[lane];
#endif
    };
    auto lookup = [&](int slot) {                               // the chunk's row ids have landed: its four rows, its Q operands
      rows[slot][0] = __builtin_amdgcn_readlane(rw[slot], 0);
      rows[slot][1] = __builtin_amdgcn_readlane(rw[slot], 31);
      rows[slot][2] = __builtin_amdgcn_readlane(rw[slot], 32);
      rows[slot][3] = __builtin_amdgcn_readlane(rw[slot], 63);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int r = hi ? rows[slot][2 * g + 1] : rows[slot][2 * g];
        // (uniform base + 32-bit byte offset: one shift-add per lookup instead of 64-bit address arithmetic;
        // launch_pass_c takes this kernel only while N * 128 fits 32 bits)
        qv[slot][g] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Q) + (((unsigned)r << 7) + koff));
      }
    };
    // one chunk: everything of slot sl0 is in registers; slot sl1's row ids have landed; slot sl2 is requested
    auto step = [&](int c, int sl0, int sl1, int sl2) {
      fetch(c + 2, sl2);
      lookup(sl1);
      float4 e4 = ev[sl0];
      if (LAZY) {
        e4.x = fmaxf(fmaf(e4.x, s1[0], t1[0]), 0.f); e4.y = fmaxf(fmaf(e4.y, s1[1], t1[1]), 0.f);
        e4.z = fmaxf(fmaf(e4.z, s1[2], t1[2]), 0.f); e4.w = fmaxf(fmaf(e4.w, s1[3], t1[3]), 0.f);
      }
      const int rwc = rw[sl0];
      if (p.logits != nullptr) {                               // classifier on this edge (mpn.py:291-292)
        if (two_class) {
          // two scalar FMA chains, kept apart: paired up by the SLP vectoriser they become v_pk_fma_f32 with op_sel,
          // the form tools/check_isa.py bans from kernels with MFMAs
          float lg0 = fmaf(cw[0][3], e4.w, fmaf(cw[0][2], e4.z, fmaf(cw[0][1], e4.y, fmaf(cw[0][0], e4.x, cbias[0]))));
          asm volatile("" : "+v"(lg0));
          const float lg1 = fmaf(cw[1][3], e4.w, fmaf(cw[1][2], e4.z, fmaf(cw[1][1], e4.y, fmaf(cw[1][0], e4.x, cbias[1]))));
          reinterpret_cast<float2*>(p.logits + (int64_t)c * 128)[lane] = make_float2(lg0, lg1);
        } else {
          const int64_t e = (int64_t)c * 64 + lane;
          for (int cc = 0; cc < p.n_classes; ++cc) {
            const float* w = p.cls_w + cc * 4;
            p.logits[e * p.n_classes + cc] = fmaf(w[3], e4.w, fmaf(w[2], e4.z, fmaf(w[1], e4.y, fmaf(w[0], e4.x, p.cls_b[cc]))));
          }
        }
      }
      // A operands: lanes 0..31 carry component k = 0 (2), lanes 32..63 component k = 1 (3) of edge (lane & 31)
      const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(e4.x), __float_as_uint(e4.y), false, false);
      const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(e4.z), __float_as_uint(e4.w), false, false);
      // membership of the chunk's edges in each group's first / last row (sorted: a prefix / a suffix of the group)
      const int ra0 = rows[sl0][0], rb0 = rows[sl0][1], ra1 = rows[sl0][2], rb1 = rows[sl0][3];
      const unsigned ma0 = (unsigned)__ballot(rwc == ra0);
      const unsigned mb0 = ra0 == rb0 ? 0u : (unsigned)__ballot(rwc == rb0);
      const unsigned ma1 = (unsigned)(__ballot(rwc == ra1) >> 32);
      const unsigned mb1 = ra1 == rb1 ? 0u : (unsigned)(__ballot(rwc == rb1) >> 32);
      const unsigned ma[2] = {ma0, ma1}, mb[2] = {mb0, mb1};
      const int ra[2] = {ra0, ra1}, rb[2] = {rb0, rb1};
      if ((ma0 | mb0) == 0xffffffffu && (ma1 | mb1) == 0xffffffffu) {
        f32x16c acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const unsigned long long member = (unsigned long long)ma[g] | ((unsigned long long)mb[g] << 32);
          const float a_c = __builtin_amdgcn_inverse_ballot_w64(member) ? 1.0f : 0.0f;
          f32x16c z = {};
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, fmaf(sk, qv[sl0][g], cb), z, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xy[g]), b0, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zw[g]), b1, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          // relu as a signed-integer max (exact for every float, no canonicalising v_max_f32 x, x, x in front); sums per
          // block of four registers = four consecutive edges of this half
          float t[16], bs[4];
#pragma unroll
          for (int i = 0; i < 16; ++i) t[i] = __int_as_float(max(__float_as_int(acc[g][i]), 0));
#pragma unroll
          for (int b = 0; b < 4; ++b) bs[b] = (t[4 * b] + t[4 * b + 1]) + (t[4 * b + 2] + t[4 * b + 3]);
          const float total = (bs[0] + bs[1]) + (bs[2] + bs[3]);
          if (mb[g] == 0u) {
            account(ra[g], total);
          } else {
            const int n_a = __builtin_popcount(ma[g]);          // 1..31 edges of the first row: a prefix
            const int q = n_a >> 3;                             // the block the boundary cuts (for one of the halves)
            const int lim = (n_a & 7) - lim_hi;                 // its registers j < lim belong to the first row
            float sa;
            auto cut = [&](int b) {
              const float c0 = 0 < lim ? t[4 * b] : 0.f, c1 = 1 < lim ? t[4 * b + 1] : 0.f;
              const float c2 = 2 < lim ? t[4 * b + 2] : 0.f, c3 = 3 < lim ? t[4 * b + 3] : 0.f;
              return (c0 + c1) + (c2 + c3);
            };
            if (q == 0) sa = cut(0);
            else if (q == 1) sa = bs[0] + cut(1);
            else if (q == 2) sa = (bs[0] + bs[1]) + cut(2);
            else sa = (bs[0] + bs[1]) + (bs[2] + cut(3));
            account(ra[g], sa);
            account(rb[g], total - sa);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          f32x16c acc = {};
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xy[g]), b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zw[g]), b1, acc, 0, 0, 0);
          slow_group(acc, rwc, g, 0xffffffffu);
        }
      }
    };
    fetch(c_beg, 0);
    fetch(c_beg + 1, 1);
    lookup(0);
    for (int c = c_beg; c < c_end; c += 3) {
      step(c, 0, 1, 2);
      if (c + 1 < c_end) step(c + 1, 1, 2, 0);
      if (c + 2 < c_end) step(c + 2, 2, 0, 1);
    }
    if (DET && n_tail && span == n_spans - 1) {
      // the partial last chunk (clamped addresses, masked passes only)
      const int64_t e = (int64_t)n_full * 64 + lane;
      const bool valid = lane < n_tail;
      const int64_t ec = valid ? e : p.n_edges - 1;
      float4 e4 = reinterpret_cast<const float4*>(p.e_out)[ec];
      const int rwc = valid ? p.row32[ec] : -1;
      if (LAZY) {
        e4.x = fmaxf(fmaf(e4.x, s1[0], t1[0]), 0.f); e4.y = fmaxf(fmaf(e4.y, s1[1], t1[1]), 0.f);
        e4.z = fmaxf(fmaf(e4.z, s1[2], t1[2]), 0.f); e4.w = fmaxf(fmaf(e4.w, s1[3], t1[3]), 0.f);
      }
      if (p.logits != nullptr && valid) {
        for (int cc = 0; cc < p.n_classes; ++cc) {
          const float* w = p.cls_w + cc * 4;
          p.logits[e * p.n_classes + cc] = fmaf(w[3], e4.w, fmaf(w[2], e4.z, fmaf(w[1], e4.y, fmaf(w[0], e4.x, p.cls_b[cc]))));
        }
      }
      const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(e4.x), __float_as_uint(e4.y), false, false);
      const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(e4.z), __float_as_uint(e4.w), false, false);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int nv = n_tail - 32 * g;
        const unsigned gm = nv >= 32 ? 0xffffffffu : (nv > 0 ? ((1u << nv) - 1u) : 0u);
        if (gm == 0) break;
        f32x16c acc = {};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(xy[g]), b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zw[g]), b1, acc, 0, 0, 0);
        slow_group(acc, rwc, g, gm);
      }
    }
    flush(true);
  }
}

__global__ __launch_bounds__(256) void classify_e0_kernel(EdgeEncParams enc, const float* attr, int64_t n_edges,
                                                          double e_total, const float* cls_w, const float* cls_b,
                                                          int n_classes, float* logits) {
  drop_resolve(enc.drop);
  __shared__ EdgeEncAffine af;
  __shared__ double scratch[kStatAttr + kStatEnc2];
  edge_enc_affine_to_smem(enc, e_total, 2, &af, scratch);
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += nthreads) {
    float a0, a1, u[4], e0[4];
    load_attr(attr, enc.fe, e, a0, a1);
    edge_enc_hidden(enc, af, e, a0, a1, u);
    edge_enc_out(enc, af, e, u, e0);
    for (int c = 0; c < n_classes; ++c) {
      float z = cls_b[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) z = fmaf(cls_w[c * 4 + j], e0[j], z);
      logits[e * n_classes + c] = z;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
#ifndef MTMC_EDGE_GRID_CAP
#define MTMC_EDGE_GRID_CAP 2048
#endif
static inline int edge_grid(int64_t n_edges, int per_block) {
  const int64_t blocks = (n_edges + per_block - 1) / per_block;
  return (int)(blocks < 1 ? 1 : (blocks > MTMC_EDGE_GRID_CAP ? MTMC_EDGE_GRID_CAP : blocks));
}

// Below kSmallEdges (kernels.h) a pass at 4 edges/thread would leave most of the 256 CUs with one or two waves: use the
// finest decomposition there (measured on camera graphs of 50k..12M edges, tools/size_sweep.py, tools/regime_sweep.py).
constexpr int64_t kSortedMaxNodes = 1ll << 24;   // pass_c_sorted_kernel keeps row ids in 24 bits
static inline int pick_ept(int64_t n_edges) { return n_edges <= kSmallEdges ? 1 : 4; }
#ifndef MTMC_PASS_A_EPT
#define MTMC_PASS_A_EPT 4
#endif
constexpr int kPassAEpt = MTMC_PASS_A_EPT;

static int size_jobs(PrepParams& p) {                    // block ranges of the passenger jobs; returns their total
  int extra = 0;
  for (int j = 0; j < p.n_jobs; ++j) {                  // ~16 float4 per lane, at most 2048 workgroups per operand
    const int64_t f4 = p.jobs[j].rows * (p.jobs[j].cols / 4), want = (f4 + 4095) / 4096;
    p.jobs[j].block0 = extra;
    p.jobs[j].n_blocks = p.jobs[j].kind == kJobSplit ? (int)((p.jobs[j].rows + 7) / 8)       // 8 rows per workgroup
                                                     : (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    extra += p.jobs[j].n_blocks;
  }
  return extra;
}
void launch_prep(const PrepParams& p0, hipStream_t s) {
  PrepParams p = p0;
  p.n_edge_blocks = p.n_edges > 0 ? edge_grid(p.n_edges, 256) : 0;
  bool any_split = false;
  for (int j = 0; j < p.n_jobs; ++j) any_split = any_split || p.jobs[j].kind == kJobSplit;
  if (any_split && p.n_edges > kSmallEdges) {           // many edges: the split jobs in a launch of their own (see above)
    PrepParams q = p0;
    q.n_edges = 0; q.n_edge_blocks = 0; q.n_jobs = 0;
    p.n_jobs = 0;
    for (int j = 0; j < p0.n_jobs; ++j) {
      if (p0.jobs[j].kind == kJobSplit) q.jobs[q.n_jobs++] = p0.jobs[j];
      else p.jobs[p.n_jobs++] = p0.jobs[j];
    }
    q.n_pass_blocks = size_jobs(q);
    hipLaunchKernelGGL(split_jobs_kernel, dim3(q.n_pass_blocks), dim3(256), 0, s, q);
    any_split = false;
  }
  p.n_pass_blocks = size_jobs(p);
  if (p.n_edge_blocks + p.n_pass_blocks == 0) return;
  if (any_split) hipLaunchKernelGGL(prep_kernel<true>, dim3(p.n_edge_blocks + p.n_pass_blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(prep_kernel<false>, dim3(p.n_edge_blocks + p.n_pass_blocks), dim3(256), 0, s, p);
}
void launch_enc2(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total, double* stat_enc2,
                 hipStream_t s) {
  hipLaunchKernelGGL(enc2_kernel, dim3(edge_grid(n_edges, 256)), dim3(256), 0, s, enc, attr, n_edges, e_total,
                     stat_enc2);
}
template <int MODE>
static void launch_pass_a_mode(const RoundParams& p, hipStream_t s) {
  // (edges per thread 1/2/4/8 x grid caps 1536..16384 swept at config 4: 0.45-0.51 ms for the three launches, flat)
  if (p.enc.drop.on) {            // training (the encoder's Dropout compiled in): graphs are small, one edge per thread
    hipLaunchKernelGGL((pass_a_kernel<1, MODE, true>), dim3(edge_grid(p.n_edges, 256)), dim3(256), 0, s, p);
    return;
  }
  switch (pick_ept(p.n_edges)) {
    case 1: hipLaunchKernelGGL((pass_a_kernel<1, MODE, false>), dim3(edge_grid(p.n_edges, 256)), dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((pass_a_kernel<kPassAEpt, MODE, false>), dim3(edge_grid(p.n_edges, 256 * kPassAEpt)), dim3(256), 0, s, p);
  }
}
// Column-blocked pass A: when, and with how many blocks.  Only where the gathered table cannot live in an XCD's L2 anyway
// (N * 16 B > 3 MB), on many-edge eval-mode lists whose sub-runs stay long enough to pay for their bookkeeping (average
// degree >= 4 per column block).  Blocks of <= 2 MB of Pc, a multiple of 8 (one residue of blockIdx % 8 per block), <= 64.
int plan_col_blocks(int64_t n_nodes, int64_t n_edges, double avg_degree, bool training) {
  const Knobs& kn = knobs();
  if (kn.no_col_blocks || training || n_edges <= kSmallEdges || n_nodes * 16 <= (int64_t)3 << 20 || n_nodes >= (1ll << 31) - 64) return 0;
  int64_t B = (n_nodes * 16 + ((int64_t)2 << 20) - 1) / ((int64_t)2 << 20);
  B = (B + 7) / 8 * 8;
  if (B > 64) B = 64;
  if (kn.col_blocks >= 8 && kn.col_blocks <= 64 && kn.col_blocks % 8 == 0) B = kn.col_blocks;
  return avg_degree >= 4.0 * (double)B ? (int)B : 0;
}
void launch_colblock_index(const RoundParams& p, int* sub, int B, int64_t row_lo, int64_t row_hi, hipStream_t s) {
  if (B <= 0 || row_hi <= row_lo) return;
  const int blk_nodes = (int)((p.n_nodes + B - 1) / B);
  const int64_t items = (row_hi - row_lo) * (B + 1), blocks = (items + 255) / 256;
  hipLaunchKernelGGL(colblock_index_kernel, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, s, p.col32, p.row_start, p.deg,
                     p.flags, row_lo, row_hi, B, blk_nodes, sub);
}
template <int MODE>
static void launch_pass_a_blocked_mode(const RoundParams& p, hipStream_t s) {
  const int B = p.col_blocks;
  const int64_t n_chunks = (p.cb_row_hi - p.cb_row_lo + 255) / 256;
  int64_t groups = 1024 / B;                         // 1024 workgroups: four per CU
  if (groups > n_chunks) groups = n_chunks;
  if (groups < 1) groups = 1;
  hipLaunchKernelGGL((pass_a_blocked_kernel<MODE>), dim3((unsigned)(groups * B)), dim3(256), 0, s, p, p.col_sub, B, p.cb_row_lo, p.cb_row_hi);
}
void launch_pass_a(const RoundParams& p0, hipStream_t s) {
  RoundParams p = p0;
  if (p.col_blocks > 0 && p.enc.drop.on) p.col_blocks = 0;
  // z1 (16 B / edge) is read by pass B, pass C and the next pass A: stored with the default policy it waits for them in the
  // 256 MB Infinity Cache; a z1 that does not fit there only pushes everything else out on its way (config 5: -2 % on the
  // edge passes with a streaming store, config 4: +2 %)
  p.stream_z1 = p.n_edges * 16 > (int64_t)256 << 20;
  if (p.col_blocks > 0) {                    // (returns at once on unsorted rows / columns; pass_a_kernel then does the round)
    switch ((p.first_round ? 1 : 0) | (p.reattach_edges ? 2 : 0)) {
      case 0: launch_pass_a_blocked_mode<0>(p, s); break;
      case 1: launch_pass_a_blocked_mode<1>(p, s); break;
      case 2: launch_pass_a_blocked_mode<2>(p, s); break;
      default: launch_pass_a_blocked_mode<3>(p, s);
    }
  }
  switch ((p.first_round ? 1 : 0) | (p.reattach_edges ? 2 : 0)) {
    case 0: launch_pass_a_mode<0>(p, s); break;
    case 1: launch_pass_a_mode<1>(p, s); break;
    case 2: launch_pass_a_mode<2>(p, s); break;
    default: launch_pass_a_mode<3>(p, s);
  }
}
void launch_pass_b(const RoundParams& p, hipStream_t s) {
  switch (pick_ept(p.n_edges)) {
    case 1: hipLaunchKernelGGL(pass_b_kernel<1>, dim3(edge_grid(p.n_edges, 256)), dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL(pass_b_kernel<4>, dim3(edge_grid(p.n_edges, 1024)), dim3(256), 0, s, p);
  }
}
// Many edges per node: the matrix-core kernel.  mfma_c = 1 (many-edge lists): it returns at once when prep_kernel found
// the rows unsorted, and pass_c_kernel, launched behind it, returns at once when they are sorted (sortedness is only known
// on the device).  mfma_c = 2 (few-edge lists, where a second launch would cost as much as the pass): the matrix-core
// kernel alone, whatever the order -- it is correct for any order, groups that touch many rows just take one masked pass
// per row.  MTMC_PASS_C_WALK=1 keeps the walk everywhere (A/B).
int plan_pass_c(int agg, bool deterministic, bool dropout, int64_t n_edges, int64_t n_nodes, double avg_degree) {
  const Knobs& kn = knobs();
  if (kn.pass_c_walk || agg == 2 || avg_degree < 24.0) return 0;
  if (dropout && (n_edges > kSmallEdges || deterministic)) return 0;     // Dropout: only the few-edge any-order kernel knows it
  // many-edge lists: the sorted kernel (also in deterministic mode) -- unless the call gets the any-order kernel instead
  // (MTMC_PASS_C_GENERAL, or >= 2^24 GLOBAL node rows, reachable with row-sharded multi-GPU calls): that one only knows
  // float atomics, so a deterministic call keeps the walk there (its carry[] is what agg_fixup_kernel adds up)
  if (n_edges > kSmallEdges) return (deterministic && (kn.pass_c_general || n_nodes >= kSortedMaxNodes)) ? 0 : 1;
  if (deterministic)                                         // fixed-order sums: the sorted kernel's DET form where it pays, else the walk
    return (n_edges > kDetSortedEdges && !kn.pass_c_general && n_nodes < kSortedMaxNodes) ? 1 : 0;
  return n_edges >= kn.pass_c_small_min ? 2 : 0;
}
void launch_seed_tick(unsigned long long* counter, unsigned long long* word, hipStream_t s) {
  hipLaunchKernelGGL(seed_tick_kernel, dim3(1), dim3(1), 0, s, counter, word);
}
int plan_edges_per_thread(int64_t n_edges) { return pick_ept(n_edges); }
bool fold_node_stat(int64_t n_edges) { return pick_ept(n_edges) == 1; }
// plan value 1 (many-edge list): pass_c_sorted_kernel, or the any-order matrix-core kernel + the walk behind it
bool pass_c_sorted_taken(int64_t n_nodes) { return !knobs().pass_c_general && n_nodes < kSortedMaxNodes; }

void launch_pass_c(const RoundParams& p0, hipStream_t s) {
  RoundParams p = p0;
  p.det_len = 32;
  p.mfma_c = plan_pass_c(p.agg, p.det != 0, p.drop_n.on != 0, p.n_edges, p.n_nodes, p.avg_degree);
  const int span_env = knobs().pass_c_span, max_blocks = knobs().pass_c_blocks;
  if (p.mfma_c == 1 && pass_c_sorted_taken(p.n_nodes)) {
    // many-edge sorted lists: a grid of twice what is resident (104 registers: four blocks per CU; the block prologue -- 74 replicated
    // statistics, one BatchNorm affine per channel -- is paid once per block) whose waves all take the same number of
    // equally long spans: the span length is chosen so that spans = waves x m for the smallest m that keeps a span at
    // <= 48 chunks (a span pays one pipeline fill; a wave that gets one span more than the others sets the kernel's time:
    // at config 4, 8-chunk spans round-robin left 6 or 7 spans per wave)
    const bool det = p.det != 0;                              // (max aggregation never gets here)
    p.mfma_c = det ? 4 : 3;                                   // (internal: 3 tells pass_c_kernel to take the tail of a sorted list,
                                                              //  4 that the deterministic kernel takes it itself)
    const int64_t n_full = p.n_edges >> 6;
    int64_t blocks = (n_full + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    const int64_t per_wave = (n_full + blocks * 4 - 1) / (blocks * 4), m = (per_wave + 47) / 48;
    const int span_c = span_env > 0 ? span_env : (int)((per_wave + (m > 0 ? m : 1) - 1) / (m > 0 ? m : 1));
    const int sc = span_c > 0 ? span_c : 1;
    p.det_len = det ? sc * 64 : 32;
    const dim3 grid((int)blocks);
    if (det) {
      if (p.lazy_e) hipLaunchKernelGGL((pass_c_sorted_kernel<true, true>), grid, dim3(256), 0, s, p, sc);
      else hipLaunchKernelGGL((pass_c_sorted_kernel<false, true>), grid, dim3(256), 0, s, p, sc);
    } else {
      if (p.lazy_e) hipLaunchKernelGGL((pass_c_sorted_kernel<true, false>), grid, dim3(256), 0, s, p, sc);
      else hipLaunchKernelGGL((pass_c_sorted_kernel<false, false>), grid, dim3(256), 0, s, p, sc);
    }
  } else if (p.mfma_c) {
    // the any-order kernel.  Many edges (lists of >= 2^24 nodes, MTMC_PASS_C_GENERAL): short spans round-robin over a
    // resident grid; few edges: one 64-edge chunk per wave, as many waves as there are chunks
    const int span_c = span_env > 0 ? span_env : (p.mfma_c == 2 ? 1 : 8);
    const int64_t spans = ((p.n_edges + 63) / 64 + span_c - 1) / span_c, blocks = (spans + 3) / 4;
    const int cap = max_blocks < 256 * 3 ? max_blocks : 256 * 3;           // 130 registers: three blocks per CU
    if (p.drop_n.on) hipLaunchKernelGGL(pass_c_mfma_kernel<true>, dim3((int)(blocks > cap ? cap : blocks)), dim3(256), 0, s, p, span_c);
    else hipLaunchKernelGGL(pass_c_mfma_kernel<false>, dim3((int)(blocks > cap ? cap : blocks)), dim3(256), 0, s, p, span_c);
  }
  // (behind a matrix-core kernel the walk returns at once on sorted lists -- all but the workgroup that owns the < 64 edges
  //  behind the last whole chunk.  Round 5 launched it with 256 workgroups instead of up to 2048 there: the launch still
  //  took 5.3 us (5.1 before: it is the fixed cost of a dependent launch, not the dispatch of 2048 workgroups that return), and
  //  an unsorted many-edge list would walk with an eighth of the workgroups -- reverted, profiles/r05_cfg4_kernel_stats.csv.
  //  With exactly ONE workgroup (-DMTMC_WALK_TAIL_GRID=1, same box, alternating): 5.60 / 5.67 us against 5.51 / 5.54 for the
  //  full grid, profiles/r05_walk_grid1.txt)
#ifdef MTMC_WALK_TAIL_GRID      // A/B builds: the walk behind the sorted kernel with this many workgroups
  if (p.mfma_c == 3 || p.mfma_c == 4) {
    hipLaunchKernelGGL(pass_c_kernel, dim3(MTMC_WALK_TAIL_GRID), dim3(kTileC), 0, s, p);
  } else
#endif
  if (p.mfma_c != 2) hipLaunchKernelGGL(pass_c_kernel, dim3(edge_grid(p.n_edges, kTileC)), dim3(kTileC), 0, s, p);
  if (p.det && p.agg != 2) {
    const int64_t blocks = (p.n_nodes + 7) / 8;
    hipLaunchKernelGGL(agg_fixup_kernel, dim3((int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, s, p);
  }
}
void launch_classify_e0(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total,
                        const float* cls_w, const float* cls_b, int n_classes, float* logits, hipStream_t s) {
  hipLaunchKernelGGL(classify_e0_kernel, dim3(edge_grid(n_edges, 256)), dim3(256), 0, s, enc, attr, n_edges, e_total,
                     cls_w, cls_b, n_classes, logits);
}

}  // namespace mtmc

#if EK_STAMP
extern "C" int mtmc_dbg_ek_stamps_edge(unsigned long long* out) {    // host buffer of 8 * 2 * 8 entries
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mtmc::g_ek), sizeof(mtmc::g_ek));
}
#endif
