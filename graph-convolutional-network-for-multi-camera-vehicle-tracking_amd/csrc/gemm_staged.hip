// Node-encoder layers 1.. of MANY-ROW graphs (reference models/mlp.py:14-27 via models/mpn.py:131): Y = relu(bn(Y_prev)) . W^T + b
// with the operand conversion taken OFF the matrix waves.
//
// gemm_bn_f16x3_kernel (gemm_bn.hip) does everything in every wave: global load -> BatchNorm affine + ReLU -> two-piece fp16
// split -> ds_write -> barrier -> ds_read -> MFMA.  Conversion and staging alternate with the matrix work instead of
// overlapping it: layer 1 of config 4 (100000 x 1024 -> 512) ran at 0.26 of its bound.  Here a 768-thread workgroup is
// split by ROLE; the waves of a workgroup go to the SIMDs cyclically (MI355X_MICROARCH.md), so every SIMD hosts one
// producer and two consumers:
//   waves 0-3  PRODUCERS  A: global load (fp32 raw Y_prev, one k-tile ahead in registers) -> affine + ReLU -> scale ->
//                            fp16 pieces (v_cvt_pkrtz) -> ds_write_b64 into the NEXT stage, in the swizzled image the
//                            fragment reads want;
//   waves 4-11 CONSUMERS  W: LDS-DMA of the layer's PRE-SPLIT weight planes (split_rows_kernel, once per forward: the same
//                            [rows][32] k-tile-major swizzled image as layer 0's operands) two k-tiles ahead, three stages;
//                         ds_read_b128 fragments + v_mfma_f32_16x16x32_f16, three products per fp32 product;
// one s_barrier per k-tile.  The VALU work of a producer runs in the issue slots the consumer's MFMAs leave free
// (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles), the weight bytes never touch a VGPR, and an A element is
// converted Nout / BN times instead of Nout / 128.
// Tile: up to 128 rows x BN = 256 columns, BK = 32 (A two stages, W three); consumers 2 x 4, 64 x BN/4 each.
// Accuracy: the same two-piece split as the other encoder kernels (22 mantissa bits per operand);
// A scale: one power of two per launch from |Y_prev|max and the BatchNorm affine (as gemm_bn_f16x3_kernel), W: one per row.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"
#include "lds_dma.h"

namespace mtmc {

#ifndef SG_STAMP
#define SG_STAMP 0        // 1: workgroup 0 records s_memtime at four points of every k-tile, per wave (tools/staged_stamps.py)
#endif
#if SG_STAMP
constexpr int kSgStampKT = 64;
__device__ unsigned long long g_sg_stamps[12 * kSgStampKT * 4];
#define SG_T(kt, pt)                                                                                          \
  do {                                                                                                        \
    if (blockIdx.x == 0 && (kt) < kSgStampKT && lane == 0) g_sg_stamps[(wid * kSgStampKT + (kt)) * 4 + (pt)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SG_T(kt, pt) do {} while (0)
#endif


constexpr int kSgBK = 32, kSgRowB = kSgBK * 2;                   // bytes per image row
constexpr int kSgNT = 768;                                       // 4 producer + 8 consumer waves: one + two per SIMD
constexpr int kSgSets = 3;                                       // register sets of the producers: A is loaded two k-tiles ahead
constexpr int kSgNW = 3;                                         // W stages: LDS-DMA runs two k-tiles ahead (a DMA takes
                                                                 // ~1 us from issue to landed under load, a k-tile less)

// The eight consumer waves form WM x WN = 2 x 4 (tile up to 128 rows x 256 columns: the wide layers) or 4 x 2 (up to 256 x 128:
// Nout = 128, where 256-column tiles do not exist and 128 x 128 ones left a wave 24 MFMAs per k-tile); either way a wave owns
// 64 rows x 64 columns = 48 MFMAs per k-tile, and the waves wm and wm + WM/2 share a SIMD.
template <int BN, int WM>
__global__ __launch_bounds__(kSgNT) void gemm_staged_kernel(StagedGemmParams p, int tiles_m, int tiles_n, int bm) {
  constexpr int WN = 8 / WM;                           // consumer waves across the tile's columns
  constexpr int BM = 64 * WM;                          // rows of the A stages (the tile uses the first bm of them)
  constexpr int NH = BM / 32;                          // rows per producer lane: 4 or 8
  constexpr int AIMG = BM * kSgRowB;                   // one A piece of one stage
  constexpr int WIMG = BN * kSgRowB;                   // one W piece of one stage
  constexpr int WC = BN / WN;                          // columns per consumer wave: 64
  constexpr int TJ = WC / 16;                          // 16-column blocks per consumer wave
  constexpr int WJ = BN / 128;                         // DMA instructions per W piece per consumer wave (128 rows per pass of the eight)
  constexpr int WST = 2 * WIMG, AST = 2 * AIMG;        // one stage of W (piece 1, piece 2) / of A
  constexpr int A0 = kSgNW * WST;                      // LDS: [kSgNW] W stages, then [2] A stages, then the input affine
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_in = reinterpret_cast<float*>(smem + A0 + 2 * AST);       // [K]
  float* t_in = s_in + p.K;                                          // [K]
  float* sc = t_in + p.K;                                            // [4]: scale of A, -, 1 / scale of A, -
  float* wred = sc + 4;                                              // [24]

  // (the integer division runs on the vector unit: pin the uniform results to scalar registers, so that everything
  // derived from the tile origin -- the LDS-DMA base addresses above all -- stays on the scalar unit)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = __builtin_amdgcn_readfirstlane((slot / tiles_n) * 8 + xcd);
  const int tn_idx = __builtin_amdgcn_readfirstlane(slot % tiles_n);
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * bm;             // bm: the tile's height (staged_tile_rows): the first WM/2 rows of consumer
  const int xr = (bm - 32 * WM) / (WM / 2);            // waves take 64 rows each, the others xr (a multiple of 16, <= 64) each
  const int n0 = tn_idx * BN;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wid < 4;

  // ---- prologue (everyone): BatchNorm affine of the K input columns, the bound on |relu(bn(.))| -> the A scale
  {
    float ms = 0.f, mt = 0.f;
    for (int kk = threadIdx.x; kk < p.K; kk += kSgNT) {
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
    if (lane == 0) { wred[wid] = ms; wred[12 + wid] = mt; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned ua = 0;
#pragma unroll
    for (int r = 0; r < kAmaxRep; ++r) ua = max(ua, p.amax_a[r]);
    float s8 = 0.f, t8 = 0.f;
#pragma unroll
    for (int w = 0; w < 12; ++w) { s8 = fmaxf(s8, wred[w]); t8 = fmaxf(t8, wred[12 + w]); }
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

  f32x4v acc[4][TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (producer) {
    // ==== PRODUCERS (waves 0-3): the A operand.  These waves issue NO LDS-DMA: every vector-memory operation of theirs is a
    // plain load the compiler counts, so its own s_waitcnt vmcnt(N) in front of the conversion is exact and leaves the
    // younger k-tiles' loads in flight.  (Mixed with hand-counted LDS-DMA in one wave the compiler's count is short by the
    // DMA instructions and its wait drains them too: the first version of this kernel paid a DMA latency per k-tile.)
    // Lane t takes the 16-byte chunk (t & 7) -- k = 4 * chunk .. + 3 of the k-tile -- of rows (t >> 3) + 32 h, h = 0..3:
    // a load instruction reads whole 128-byte lines, eight rows per wave.
    const int pt = threadIdx.x, c8 = pt & 7, r0 = pt >> 3;
    const float* a_src[NH];
    unsigned a_dst[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int r = r0 + 32 * h;
      const int64_t row = (r < bm && m0 + r < p.M) ? m0 + r : m0;    // rows past M or past the tile's height: the tile's
      a_src[h] = p.A + row * p.lda + c8 * 4;                         // first row again (cached; what they feed is never stored)
      a_dst[h] = (unsigned)(r * kSgRowB + (((c8 >> 1) ^ plane_swz(r)) << 4) + (c8 & 1) * 8);   // the fragment reads' swizzle
    }
    float4 ra[kSgSets][NH];                              // [register set][row]
    auto load_a = [&](int kt, int set) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        ra[set][h] = *reinterpret_cast<const float4*>(a_src[h] + kt * kSgBK);
      }
    };
    auto convert_a = [&](int kt, int set) {
      unsigned char* st = smem + A0 + (kt & 1) * AST;
      const float4 s4 = *reinterpret_cast<const float4*>(s_in + kt * kSgBK + c8 * 4);
      const float4 t4 = *reinterpret_cast<const float4*>(t_in + kt * kSgBK + c8 * 4);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float4 v = ra[set][h];
        // relu(bn(y)) scaled into fp16's range, then h1 = rtz(x), h2 = rtz(x - h1) (exact residual; see gemm_bn.hip `put`).
        // (Folding the scale into s4 / t4 and one LDS address per row -- six instructions per float4 less -- measured
        // 3 % SLOWER, 368 against 358 us on layer 1 of config 4, same box: kept as it was.)
        const float x0 = fmaxf(fmaf(v.x, s4.x, t4.x), 0.f) * sa, x1 = fmaxf(fmaf(v.y, s4.y, t4.y), 0.f) * sa;
        const float x2 = fmaxf(fmaf(v.z, s4.z, t4.z), 0.f) * sa, x3 = fmaxf(fmaf(v.w, s4.w, t4.w), 0.f) * sa;
        const h2_t a01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), a23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
        const h2_t b01 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)a01[0], x1 - (float)a01[1]);
        const h2_t b23 = __builtin_amdgcn_cvt_pkrtz(x2 - (float)a23[0], x3 - (float)a23[1]);
        uint2 q1, q2;
        q1.x = __builtin_bit_cast(unsigned, a01); q1.y = __builtin_bit_cast(unsigned, a23);
        q2.x = __builtin_bit_cast(unsigned, b01); q2.y = __builtin_bit_cast(unsigned, b23);
        *reinterpret_cast<uint2*>(st + a_dst[h]) = q1;
        *reinterpret_cast<uint2*>(st + AIMG + a_dst[h]) = q2;
      }
    };
    // One k-tile of producer work.  A(k) lives in register set k % 3: tile kt+3 is loaded into set `ld` = kt % 3 (free:
    // A(kt) went to LDS one k-tile ago) while tile kt+1, loaded TWO k-tiles ago, is converted out of set `cv` = (kt+1) % 3.
    // ld / cv are literals at the call sites, so the register arrays are indexed statically after inlining.
    auto step = [&](int kt, int ld, int cv) {
      SG_T(kt, 0);
      if (kt + 3 < nk) load_a(kt + 3, ld);
      if (kt + 1 < nk) convert_a(kt + 1, cv);            // -> A stage (kt+1)&1: the consumers left it at the last barrier
      SG_T(kt, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SG_T(kt, 2);
      __builtin_amdgcn_s_barrier();
      SG_T(kt, 3);
    };
    load_a(0, 0);
    if (nk > 1) load_a(1, 1);
    if (nk > 2) load_a(2, 2);
    convert_a(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // pipeline filled (the consumers' W(0) has landed too)
    for (int kt = 0; kt < nk; kt += 3) {
      step(kt, 0, 1);
      if (kt + 1 < nk) step(kt + 1, 1, 2);
      if (kt + 2 < nk) step(kt + 2, 2, 0);
    }
  } else {
    // ==== CONSUMERS (waves 4-11, two per SIMD beside one producer): 2 x 4 waves, 64 rows x WC columns each, as 4 x TJ
    // blocks of v_mfma_f32_16x16x32_f16 (the layer-0 kernel's fragment reads, gemm_presplit.hip).  They also bring in the
    // W operand -- LDS-DMA of the pre-split weight planes, one k-tile ahead, their only vector-memory traffic, so the
    // waits below are exact; while one wave of a SIMD issues its DMA or waits for fragments, its partner's MFMAs keep the
    // matrix pipe busy.  Thread t fills chunk (t & 3) of rows (t >> 2) + 128 j of both pieces: a uniform base (SGPRs,
    // advanced by SALU) + a loop-invariant lane offset.
    const int cw = wid - 4, ct = threadIdx.x - 256;
    const int wm = cw / WN, wn = cw % WN;
    unsigned off_w[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int r = (ct >> 2) + 128 * j;
      const int br = n0 + r < p.Nout ? r : p.Nout - 1 - n0;
      off_w[j] = (unsigned)(br * kSgBK + (ct & 3) * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const char* w_tile = reinterpret_cast<const char*>(p.Wh + (int64_t)n0 * kSgBK);
    const int64_t w_plane = (int64_t)p.Nout * p.K * 2, w_kt = (int64_t)p.Nout * kSgBK * 2;
    // one of the wave's 2 * WJ LDS-DMA instructions of k-tile kt (g = piece * WJ + pass)
    auto dma_one = [&](int kt, int g) {
      const int q = g / WJ, j = g % WJ;
      lds_dma16(w_tile + q * w_plane + kt * w_kt, off_w[j], lds0 + (kt % kSgNW) * WST + cw * 1024 + q * WIMG + j * 8192);
    };
    auto dma_w = [&](int kt) {
#pragma unroll
      for (int g = 0; g < 2 * WJ; ++g) dma_one(kt, g);
    };
    const int r16 = lane & 15, ks = lane >> 4;                       // fragment row inside a 16-row block, 8-half k group
    const int so = (ks ^ plane_swz(r16)) * 16;                       // the stored swizzle (lds_dma.h)
    const int wrow = wm < WM / 2 ? 64 * wm : 32 * WM + (wm - WM / 2) * xr;    // this wave's first row of the tile
    const int a_row = (wrow + r16) * kSgRowB + so, b_row = (wn * WC + r16) * kSgRowB + so;
    const int nblk = wm < WM / 2 ? 4 : xr / 16;                      // 16-row blocks of this wave

    dma_w(0);
    if (nk > 1) {
      dma_w(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WJ) : "memory");  // W(0) landed, W(1) may fly
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // pairs with the producers' pipeline-fill barrier
    for (int kt = 0; kt < nk; ++kt) {
      SG_T(kt, 0);
      const bool more = kt + 2 < nk;                     // W(kt+2) -> W stage (kt+2)%3 = (kt-1)%3: everyone left it at the last barrier
      const unsigned char* st = smem + (kt % kSgNW) * WST;
      const unsigned char* as = smem + A0 + (kt & 1) * AST;
      f16x8 b[TJ][2];
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          b[j][q] = *reinterpret_cast<const f16x8*>(st + q * WIMG + b_row + j * 16 * kSgRowB);
        }
      // the A fragments of block i + 1 are read BEFORE block i's MFMAs (two register sets): read right before their use,
      // every block began with an exposed LDS round trip
      f16x8 af[2][2];
      auto read_a = [&](int i, int set) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          af[set][q] = *reinterpret_cast<const f16x8*>(as + q * AIMG + a_row + i * 16 * kSgRowB);
        }
      };
      read_a(0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) read_a(i + 1, (i + 1) & 1);
        const f16x8* a = af[i & 1];
        if (i < nblk) {                                    // (rows past the tile's height: wave-uniform skip)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {                   // the three products of a block back to back (gemm_presplit.hip)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j][1], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          }
        }
        // the next k-tile's LDS-DMA goes out BETWEEN the blocks, behind MFMAs that are already queued: issued in one run
        // right after the barrier, by all eight waves at once, it kept the matrix pipes idle for the CU's whole address queue
        if (more) {
#pragma unroll
          for (int g = i * (2 * WJ) / 4; g < (i + 1) * (2 * WJ) / 4; ++g) dma_one(kt + 2, g);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // this wave's fragment reads are done (the stages may be refilled), and its share of W(kt+1) has landed -- W(kt+2),
      // issued during this k-tile, stays in flight across the barrier
      SG_T(kt, 1);
      if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * WJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      SG_T(kt, 2);
      __builtin_amdgcn_s_barrier();
      SG_T(kt, 3);
    }
  }

  // ---- epilogue (consumers; the producers only join the barriers): undo the scales, bias, raw Y through a per-wave LDS
  // transposition (whole row segments of 4 * WC bytes, as gemm_f16p_m16_kernel), fp64 column statistics, |Y|max
  __syncthreads();
  const int cw = (wid + 4) & 7, wm = cw / WN, wn = cw % WN;           // (wid - 4 for the consumers)
  float* stg = reinterpret_cast<float*>(smem) + cw * (16 * WC);
  double* colred = reinterpret_cast<double*>(smem + 8 * 16 * WC * sizeof(float));   // [WM][2 (sum, sq)][BN], behind the strips
  float ymax = 0.f;
  if (!producer) {
    const int r16 = lane & 15, ks = lane >> 4;
    const float inv_a = sc[2];
    const bool vec_ok = (p.ldy & 3) == 0 && ((uintptr_t)p.Y & 15) == 0;
    float bias[TJ], iw[TJ];
    double cs[TJ], cq[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * WC + j * 16 + r16;
      const bool cok = col < p.Nout;
      bias[j] = cok ? p.bias[col] : 0.f;
      iw[j] = cok ? p.inv_w[col] : 0.f;
      cs[j] = cq[j] = 0;
    }
    constexpr int LPR = WC / 4;                          // lanes per staged row on the way out (16 bytes each): 16 or 8
    constexpr int RPI = 64 / LPR;                        // rows per store instruction: 4 or 8
    const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
    const int gcol = n0 + wn * WC + rcol;
    const int nblk = wm < WM / 2 ? 4 : xr / 16;
    const int wrow = wm < WM / 2 ? 64 * wm : 32 * WM + (wm - WM / 2) * xr;
    const int64_t m_end = m0 + bm < p.M ? m0 + bm : p.M;             // rows of this tile: [m0, m_end)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nblk) {                                       // (wave-uniform; no `break`: the loop must unroll, acc[] is registers)
      const int64_t row0 = m0 + wrow + i * 16;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const bool cok = n0 + wn * WC + j * 16 + r16 < p.Nout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = fmaf(acc[i][j][r] * inv_a, iw[j], bias[j]);
          stg[(4 * ks + r) * WC + j * 16 + r16] = y;
          if (row0 + 4 * ks + r < m_end && cok) {
            ymax = fmaxf(ymax, fabsf(y));
            cs[j] += y;
            cq[j] += (double)y * y;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 16 / RPI; ++q) {
        const int rr = rrow + RPI * q;
        const int64_t row = row0 + rr;
        const float4 v = *reinterpret_cast<const float4*>(stg + rr * WC + rcol);
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
    for (int j = 0; j < TJ; ++j) {
      const int cl = wn * WC + j * 16 + r16;
      double a = cs[j], b = cq[j];
      a += __shfl_xor(a, 16, 64);
      b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 32, 64);
      if (lane < 16) {
        colred[(wm * 2 + 0) * BN + cl] = a;
        colred[(wm * 2 + 1) * BN + cl] = b;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * BN; i += kSgNT) {
    const int which = i / BN, cl = i % BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out) {
      double v = 0;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += colred[(w * 2 + which) * BN + cl];
      unsafeAtomicAdd(p.stats_out + which * p.Nout + col, v);
    }
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
      for (int w = 1; w < 12; ++w) m = fmaxf(m, wmax[w]);
      atomicMax(p.amax_y + (blockIdx.x % kAmaxRep), __float_as_uint(m));
    }
  }
}

// which layers: eval mode (no Dropout here), an input BatchNorm (layers >= 1), many rows (the plan of the in-loop kernel
// would be 128 x 128 tiles), K a multiple of 32 whose affine fits beside the stages, and
//   Nout a multiple of 256: 128-row x 256-column tiles (consumers 2 x 4);
//   Nout = 128 and enough rows to fill the chip with 256-row tiles: 256 x 128 (consumers 4 x 2).  (128 x 128 tiles with 2 x 4
//   consumers -- 24 MFMAs per wave and k-tile -- measured 0.122 ms on 100000 x 512 -> 128 against 0.094 ms in-loop.)
static int staged_wm(int64_t rows, int Nout) {           // rows of consumer waves: 2, 4, or 0 = not a role-split layer
  if (Nout >= 256 && Nout % 256 == 0) return rows >= 4096 ? 2 : 0;
  if (Nout == 128) return rows >= 49152 ? 4 : 0;
  return 0;
}

bool staged_layer(int64_t rows, int K, int Nout) {
  const Knobs& kn = knobs();
  if (kn.gemm_no_staged || kn.gemm_fp32 || kn.gemm_no_f16) return false;
  int sk;
  return staged_wm(rows, Nout) != 0 && K % 32 == 0 && K >= 64 && K <= 2048 && gemm_plan(rows, K, Nout, &sk) == 2;
}

// Tile height: with M / 128 x Nout / BN tiles on 256 CUs the last round of workgroups can be nearly empty (config 4, layer 1:
// 1564 tiles = 6.1 rounds, paid as 7).  The height is a launch parameter: the first wm/2 rows of consumer waves take 64 rows
// each, the others xr each (a multiple of 16 up to 64), and the two waves that share a SIMD always add up to 64 + xr -- heights
// 80..128 step 16 for wm = 2, 160..256 step 32 for wm = 4.  Take the one that minimises rounds x height: 112 rows for that
// layer 1 (1786 tiles = 6.98 rounds of 7/8 of the work each), 224 for its layer 2 (447 tiles, two rounds).
int staged_tile_rows(int64_t M, int tiles_n, int wm) {
  const int forced = knobs().staged_xr;                   // MTMC_STAGED_XR: rows of the second half of the consumer waves
  if (forced >= 16 && forced <= 64 && forced % 16 == 0) return 32 * wm + (wm / 2) * forced;
  int best = 64 * wm;
  int64_t best_cost = -1;
  for (int xr = 64; xr >= 16; xr -= 16) {
    const int bm = 32 * wm + (wm / 2) * xr;
    const int64_t tiles = (M + bm - 1) / bm * tiles_n, rounds = (tiles + 255) / 256;
    // + 12 wm: a tile's fixed cost (its W stream, prologue, epilogue) in row equivalents, fitted on one box (MTMC_STAGED_XR):
    // layer 1 at 1M / 125k / 100k rows took 4.17 / 0.603 / 0.508 ms with 128-row tiles and 4.24 / 0.621 / 0.496 with 112
    const int64_t cost = rounds * (bm + 12 * wm);
    if (best_cost < 0 || cost < best_cost) { best = bm; best_cost = cost; }
  }
  return best;
}

template <int BN, int WM>
static int launch_staged(const StagedGemmParams& p, hipStream_t s) {
  const int tiles_n = p.Nout / BN;
  const int bm = staged_tile_rows(p.M, tiles_n, WM);
  const int tiles_m = (int)((p.M + bm - 1) / bm);
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)kSgNW * 2 * BN * kSgRowB + (size_t)2 * 2 * (64 * WM) * kSgRowB + (size_t)(2 * p.K + 4 + 24) * sizeof(float);
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_staged_kernel<BN, WM>), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL((gemm_staged_kernel<BN, WM>), dim3(grid), dim3(kSgNT), lds, s, p, tiles_m, tiles_n, bm);
  return MTMC_OK;
}

int launch_gemm_staged(const StagedGemmParams& p, hipStream_t s) {
  if (p.M < 1 || p.K % 32 || p.K < 64 || p.K > 2048 || !p.stats_in || !p.amax_a) return 1;
  if (p.Nout >= 256 && p.Nout % 256 == 0) return launch_staged<256, 2>(p, s);
  if (p.Nout == 128) return launch_staged<128, 4>(p, s);
  return 1;
}

}  // namespace mtmc

#if SG_STAMP
extern "C" int mtmc_dbg_staged_stamps(unsigned long long* out) {    // host buffer of 12 * 64 * 4 entries
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mtmc::g_sg_stamps), sizeof(mtmc::g_sg_stamps));
}
#endif
