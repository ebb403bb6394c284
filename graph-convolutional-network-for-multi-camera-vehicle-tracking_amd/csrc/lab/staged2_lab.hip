// Node-encoder layer 1 of MANY-ROW graphs, second form of the role-split kernel (gemm_staged.hip): the W operand does not
// pass through LDS at all, and the twelve waves meet once per TWO k-tiles.
//
// gemm_staged_kernel<256, 2> spends 2930 cycles per 32-column k-tile against 1344 of matrix-pipe work per SIMD (DESIGN.md
// 3.1, the s_memtime timeline): every k-tile ends in a rendezvous of twelve waves, and each consumer wave issues four
// LDS-DMA instructions per k-tile (60-185 cycles apiece beside MFMAs and fragment reads) and eight B-fragment reads on top of
// its eight A-fragment reads.  A BK = 64 form of that kernel needs 192 KB of LDS.  Here:
//   * the eight consumer waves form 1 x 8: a wave owns ALL rows of the tile (up to 128) and 32 columns, 8 x 2 blocks of
//     v_mfma_f32_16x16x32_f16 = the same 48 MFMAs per k-tile, the same 64 accumulator registers;
//   * so a k-tile's W data is read by exactly ONE wave of the workgroup -- it can come straight from the pre-split planes in
//     L2 into that wave's registers: the planes are k-tile-major and hold the swizzled fragment image (lds_dma.h), a wave's
//     (16 columns x 32 k, one piece) fragment is one coalesced 1 KB global_load_dwordx4, four per k-tile, requested one
//     k-tile ahead into a second register set.  No LDS-DMA, no W stages, no B-fragment reads; the same 32 KB per k-tile
//     leave the L2 as before.  Every vector-memory operation of every wave is a plain load or store the compiler counts;
//   * LDS holds only A: FOUR stages of 16 KB.  Interval t = k-tiles 2t, 2t+1: the producers convert tiles 2t+2, 2t+3 into
//     the stages the consumers left at the last barrier while the consumers multiply tiles 2t, 2t+1 -- one s_barrier per 64
//     columns of K instead of one per 32;
//   * the A fragments of a wave's eight row blocks are read two blocks ahead (three register sets): six MFMAs cover a
//     fragment round trip, twelve two.
// Same arithmetic, same accuracy, same epilogue (per-wave LDS transposition, fp64 column statistics, |Y|max) as
// gemm_staged_kernel.  Nout a multiple of 256, K a multiple of 64.
//
// KERNEL LABORATORY (libmtmc_lab.so), not the product: measured on one box against gemm_staged_kernel<256,2>
// (profiles/r04_staged2_ab.txt) it takes the SAME time -- layer 1 of config 4 0.353 / 0.361 ms against 0.364 / 0.365, 1M rows
// 3.97 against 4.00 ms, 100000 x 2048 -> 1024 1.43 against 1.43 ms -- although it halves the rendezvous, issues no LDS-DMA and
// reads no B fragments.  What bounds the role-split form is therefore not the consumers' instruction stream or the barrier
// count.  Entry point: mtmc_lab_linear_staged2_raw (tools/staged2_ab.sh, tests/test_gpu_gemm_staged.py).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "../kernels.h"
#include "../lds_dma.h"

namespace mtmc {

constexpr int kS2BK = 32, kS2RowB = kS2BK * 2;        // bytes per image row of a k-tile
constexpr int kS2NT = 768;                            // 4 producer + 8 consumer waves
constexpr int kS2BM = 128, kS2BN = 256;               // rows of an A stage / columns of a tile
constexpr int kS2Stages = 4;                          // A stages: two being read, two being written
constexpr int kS2AIMG = kS2BM * kS2RowB;              // one piece of one A stage (8 KB)
constexpr int kS2AST = 2 * kS2AIMG;                   // one A stage (16 KB)
#ifndef S2_DEEP
#define S2_DEEP 0                                     // 1: six producer register sets, A requested two intervals (four k-tiles) ahead
#endif
constexpr int kS2Sets = S2_DEEP ? 6 : 4;              // producer register sets: A is loaded up to three (five) k-tiles ahead

__global__ __launch_bounds__(kS2NT) void gemm_staged_w_kernel(StagedGemmParams p, int tiles_m, int tiles_n, int bm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_in = reinterpret_cast<float*>(smem + kS2Stages * kS2AST);   // [K]
  float* t_in = s_in + p.K;                                            // [K]
  float* sc = t_in + p.K;                                              // [4]: scale of A, -, 1 / scale of A, -
  float* wred = sc + 4;                                                // [24]

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tm_idx = __builtin_amdgcn_readfirstlane((slot / tiles_n) * 8 + xcd);
  const int tn_idx = __builtin_amdgcn_readfirstlane(slot % tiles_n);
  if (tm_idx >= tiles_m) return;
  const int64_t m0 = (int64_t)tm_idx * bm;
  const int n0 = tn_idx * kS2BN;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wid < 4;
  const int nblk = bm / 16;                             // 16-row blocks of the tile (5 .. 8)

  // ---- prologue (everyone): BatchNorm affine of the K input columns, the bound on |relu(bn(.))| -> the A scale
  {
    float ms = 0.f, mt = 0.f;
    for (int kk = threadIdx.x; kk < p.K; kk += kS2NT) {
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
  const int nk = p.K / kS2BK;                           // even (K % 64 == 0)
  const int n_iv = nk / 2;                              // barrier intervals

  f32x4v acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (producer) {
    // ==== PRODUCERS (waves 0-3): lane t takes the 16-byte chunk (t & 7) of rows (t >> 3) + 32 h, h = 0..3, of every k-tile:
    // plain global loads of the raw fp32 activations up to three k-tiles ahead (four register sets) -> BatchNorm affine +
    // ReLU -> scale -> two fp16 pieces -> ds_write_b64 into stage kt % 4 in the fragment reads' swizzle.
    const int pt = threadIdx.x, c8 = pt & 7, r0 = pt >> 3;
    const float* a_src[4];
    unsigned a_dst[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = r0 + 32 * h;
      const int64_t row = (r < bm && m0 + r < p.M) ? m0 + r : m0;    // rows past M or the tile's height: the tile's first row
      a_src[h] = p.A + row * p.lda + c8 * 4;                         // again (cached; what they feed is never stored)
      a_dst[h] = (unsigned)(r * kS2RowB + (((c8 >> 1) ^ plane_swz(r)) << 4) + (c8 & 1) * 8);
    }
    float4 ra[kS2Sets][4];
    auto load_a = [&](int kt, int set) {
#pragma unroll
      for (int h = 0; h < 4; ++h) ra[set][h] = *reinterpret_cast<const float4*>(a_src[h] + kt * kS2BK);
    };
    auto convert_a = [&](int kt, int set) {
      unsigned char* st = smem + (kt % kS2Stages) * kS2AST;
      const float4 s4 = *reinterpret_cast<const float4*>(s_in + kt * kS2BK + c8 * 4);
      const float4 t4 = *reinterpret_cast<const float4*>(t_in + kt * kS2BK + c8 * 4);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 v = ra[set][h];
        const float x0 = fmaxf(fmaf(v.x, s4.x, t4.x), 0.f) * sa, x1 = fmaxf(fmaf(v.y, s4.y, t4.y), 0.f) * sa;
        const float x2 = fmaxf(fmaf(v.z, s4.z, t4.z), 0.f) * sa, x3 = fmaxf(fmaf(v.w, s4.w, t4.w), 0.f) * sa;
        const h2_t a01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), a23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
        const h2_t b01 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)a01[0], x1 - (float)a01[1]);
        const h2_t b23 = __builtin_amdgcn_cvt_pkrtz(x2 - (float)a23[0], x3 - (float)a23[1]);
        uint2 q1, q2;
        q1.x = __builtin_bit_cast(unsigned, a01); q1.y = __builtin_bit_cast(unsigned, a23);
        q2.x = __builtin_bit_cast(unsigned, b01); q2.y = __builtin_bit_cast(unsigned, b23);
        *reinterpret_cast<uint2*>(st + a_dst[h]) = q1;
        *reinterpret_cast<uint2*>(st + kS2AIMG + a_dst[h]) = q2;
      }
    };
    // A(k) lives in register set k % 4.  Before the loop: tiles 0..3 requested, tiles 0 and 1 converted (stages 0, 1).
    // Interval t: tiles 2t+2, 2t+3 are converted out of sets (2t+2) % 4, (2t+3) % 4 -- loaded one interval ago -- into stages
    // (2t+2) % 4, (2t+3) % 4 (the consumers left them at the last barrier), and tiles 2t+4, 2t+5 are requested into the two
    // sets tiles 2t, 2t+1 occupied.  Set indices are literals at the call sites (the arrays stay in registers).
    auto interval = [&](int t, int s_cv0, int s_cv1, int s_ld0, int s_ld1) {
      const int kt = 2 * t;
      if (kt + 4 < nk) load_a(kt + 4, s_ld0);
      if (kt + 5 < nk) load_a(kt + 5, s_ld1);
      if (kt + 2 < nk) convert_a(kt + 2, s_cv0);
      if (kt + 3 < nk) convert_a(kt + 3, s_cv1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
#if S2_DEEP
    // six sets, tile k in set k % 6: interval t converts tiles 2t+2, 2t+3 and requests tiles 2t+6, 2t+7 (into the sets of
    // tiles 2t, 2t+1): every load has two intervals to arrive
    auto interval6 = [&](int t, int s_cv0, int s_cv1, int s_ld0, int s_ld1) {
      const int kt = 2 * t;
      if (kt + 6 < nk) load_a(kt + 6, s_ld0);
      if (kt + 7 < nk) load_a(kt + 7, s_ld1);
      if (kt + 2 < nk) convert_a(kt + 2, s_cv0);
      if (kt + 3 < nk) convert_a(kt + 3, s_cv1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    load_a(0, 0);
    load_a(1, 1);
    if (nk > 2) load_a(2, 2);
    if (nk > 3) load_a(3, 3);
    if (nk > 4) load_a(4, 4);
    if (nk > 5) load_a(5, 5);
    convert_a(0, 0);
    convert_a(1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // pipeline filled
    for (int t = 0; t < n_iv; t += 3) {
      interval6(t, 2, 3, 0, 1);
      if (t + 1 < n_iv) interval6(t + 1, 4, 5, 2, 3);
      if (t + 2 < n_iv) interval6(t + 2, 0, 1, 4, 5);
    }
#else
    load_a(0, 0);
    load_a(1, 1);
    if (nk > 2) load_a(2, 2);
    if (nk > 3) load_a(3, 3);
    convert_a(0, 0);
    convert_a(1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // pipeline filled
    for (int t = 0; t < n_iv; t += 2) {
      interval(t, 2, 3, 0, 1);                           // even interval: converts tiles 4u+2, 4u+3, loads into sets 0, 1
      if (t + 1 < n_iv) interval(t + 1, 0, 1, 2, 3);     // odd interval: converts tiles 4u+4, 4u+5 (sets 0, 1), loads into 2, 3
    }
#endif
  } else {
    // ==== CONSUMERS (waves 4-11, two per SIMD beside one producer): 1 x 8, wave cw owns columns n0 + 32 cw .. + 31 and all
    // rows.  W fragments: global -> registers, one k-tile ahead; A fragments: LDS, two blocks ahead.
    const int cw = wid - 4;
    const int r16 = lane & 15, ks = lane >> 4;                       // fragment row inside a 16-row block, 8-half k group
    const int so = (ks ^ plane_swz(r16)) * 16;                       // the stored swizzle (lds_dma.h)
    const int a_row = r16 * kS2RowB + so;
    // this lane's 16 bytes of the (column block j, piece q) fragment of k-tile kt: planes are [piece][k-tile][Nout][32]
    const int64_t w_plane = (int64_t)p.Nout * p.K * 2, w_kt = (int64_t)p.Nout * kS2BK * 2;
    const char* w_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + cw * 32 + j * 16 + r16;
      col = col < p.Nout ? col : p.Nout - 1;
      w_lane[j] = reinterpret_cast<const char*>(p.Wh) + (int64_t)col * kS2RowB + so;
    }
    f16x8 bq[2][2][2];                                               // [set][column block][piece]
    auto load_w = [&](int kt, int set) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) bq[set][j][q] = *reinterpret_cast<const f16x8*>(w_lane[j] + q * w_plane + kt * w_kt);
    };
    // one k-tile: 8 row blocks x (2 column blocks x 3 products); the A fragments of block i + 2 are read before block i's MFMAs
    auto ktile = [&](int kt, int set) {
      const unsigned char* as = smem + (kt % kS2Stages) * kS2AST + a_row;
      f16x8 af[3][2];
      auto read_a = [&](int i, int s3) {
#pragma unroll
        for (int q = 0; q < 2; ++q) af[s3][q] = *reinterpret_cast<const f16x8*>(as + q * kS2AIMG + i * 16 * kS2RowB);
      };
      read_a(0, 0);
      read_a(1, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i + 2 < 8) read_a(i + 2, (i + 2) % 3);
        const f16x8* a = af[i % 3];
        if (i < nblk) {                                      // (row blocks past the tile's height: wave-uniform skip)
#pragma unroll
          for (int j = 0; j < 2; ++j) {                      // the three products of a block back to back (gemm_presplit.hip)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bq[set][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[set][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[set][j][1], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          }
        }
      }
    };
    load_w(0, 0);
    __builtin_amdgcn_s_barrier();                        // pairs with the producers' pipeline-fill barrier: A(0), A(1) written
    for (int t = 0; t < n_iv; ++t) {
      const int kt = 2 * t;
      load_w(kt + 1, 1);                                 // (nk is even: tile kt + 1 exists)
      ktile(kt, 0);
      if (kt + 2 < nk) load_w(kt + 2, 0);
      ktile(kt + 1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragment reads are done: stages kt % 4, (kt + 1) % 4 are free
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue (consumers; the producers only join the barriers): undo the scales, bias, raw Y through a per-wave LDS
  // transposition (whole 128-byte row segments), fp64 column statistics, |Y|max
  __syncthreads();
  constexpr int WC = 32;
  const int cw = (wid + 4) & 7;                                        // (wid - 4 for the consumers)
  float* stg = reinterpret_cast<float*>(smem) + cw * (16 * WC);
  double* colred = reinterpret_cast<double*>(smem + 8 * 16 * WC * sizeof(float));   // [2 (sum, sq)][BN], behind the strips
  float ymax = 0.f;
  if (!producer) {
    const int r16 = lane & 15, ks = lane >> 4;
    const float inv_a = sc[2];
    const bool vec_ok = (p.ldy & 3) == 0 && ((uintptr_t)p.Y & 15) == 0;
    float bias[2], iw[2];
    double cs[2], cq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + cw * WC + j * 16 + r16;
      const bool cok = col < p.Nout;
      bias[j] = cok ? p.bias[col] : 0.f;
      iw[j] = cok ? p.inv_w[col] : 0.f;
      cs[j] = cq[j] = 0;
    }
    constexpr int LPR = WC / 4, RPI = 64 / LPR;          // 8 lanes per staged row on the way out, 8 rows per store instruction
    const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
    const int gcol = n0 + cw * WC + rcol;
    const int64_t m_end = m0 + bm < p.M ? m0 + bm : p.M;             // rows of this tile: [m0, m_end)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < nblk) {                                     // (wave-uniform; no `break`: the loop must unroll, acc[] is registers)
        const int64_t row0 = m0 + i * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool cok = n0 + cw * WC + j * 16 + r16 < p.Nout;
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
    for (int j = 0; j < 2; ++j) {
      const int cl = cw * WC + j * 16 + r16;
      double a = cs[j], b = cq[j];
      a += __shfl_xor(a, 16, 64);
      b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 32, 64);
      if (lane < 16) {
        colred[cl] = a;
        colred[kS2BN + cl] = b;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * kS2BN; i += kS2NT) {
    const int which = i / kS2BN, cl = i % kS2BN, col = n0 + cl;
    if (col < p.Nout && p.stats_out) unsafeAtomicAdd(p.stats_out + which * p.Nout + col, colred[which * kS2BN + cl]);
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

int launch_gemm_staged2(const StagedGemmParams& p, hipStream_t s) {
  if (p.M < 1 || p.Nout < 256 || p.Nout % 256 || p.K % 64 || p.K < 128 || p.K > 2048 || !p.stats_in || !p.amax_a) return 1;
  const int tiles_n = p.Nout / kS2BN;
  const int bm = staged_tile_rows(p.M, tiles_n, 2);     // 80 .. 128 rows, the height that fills the last round of workgroups
  const int tiles_m = (int)((p.M + bm - 1) / bm);
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  const size_t lds = (size_t)kS2Stages * kS2AST + (size_t)(2 * p.K + 4 + 24) * sizeof(float);
  if (!allow_big_lds(reinterpret_cast<const void*>(gemm_staged_w_kernel), 160 * 1024)) return MTMC_E_HIP;
  hipLaunchKernelGGL(gemm_staged_w_kernel, dim3(grid), dim3(kS2NT), lds, s, p, tiles_m, tiles_n, bm);
  return MTMC_OK;
}

}  // namespace mtmc

// The laboratory twin of mtmc_linear_staged_raw (csrc/api.hip): same arguments, the second form of the kernel.
extern "C" int32_t mtmc_lab_linear_staged2_raw(const float* A, int64_t lda, const double* stats_in, const float* gamma_in,
                                               const float* beta_in, double count, const float* W, const float* bias, float* Y,
                                               int64_t M, int32_t K, int32_t N, void* work, uint64_t work_bytes, uint32_t* scratch,
                                               double* stats, void* stream) {
  if (!A || !stats_in || !gamma_in || !beta_in || !W || !bias || !Y || !work || !scratch || M < 1 || lda < K || (lda & 3) ||
      ((uintptr_t)A & 15))
    return MTMC_E_ARG;
  const uint64_t iw_off = ((uint64_t)N * K * 4 + 255) / 256 * 256;
  if (work_bytes < iw_off + (uint64_t)N * 4) return MTMC_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, 3 * mtmc::kAmaxRep * sizeof(uint32_t), s) != hipSuccess) return MTMC_E_HIP;
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return MTMC_E_HIP;
  mtmc::PrepParams pp = {};
  pp.n_edges = 0; pp.n_jobs = 1;
  pp.jobs[0] = {A, M, K, lda, scratch, 0, 0};
  mtmc::launch_prep(pp, s);
  unsigned char* wk = static_cast<unsigned char*>(work);
  mtmc::launch_split_rows(W, K, N, K, wk, reinterpret_cast<float*>(wk + iw_off), s);
  mtmc::StagedGemmParams g;
  g.A = A; g.lda = lda; g.stats_in = stats_in; g.gamma_in = gamma_in; g.beta_in = beta_in; g.count = count;
  g.amax_a = scratch; g.Wh = reinterpret_cast<const _Float16*>(wk); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
  g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
  g.M = M; g.K = K; g.Nout = N;
  const int rc = mtmc::launch_gemm_staged2(g, s);
  if (rc != 0) return rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG;
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}
