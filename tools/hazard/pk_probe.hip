// Stand-alone probe for the packed-fp32 failure found in gemm_bn_f16x3_kernel's RNE build:
//     v_pk_mul_f32 / v_pk_fma_f32  D, A, S  op_sel:[0,1(,0)]      (the LOW result lane reads the HIGH dword of S)
// delivered 0 in the low result lane of lanes 48-63, sporadically, once two workgroups shared a CU.
// The probe runs the same instruction forms inside a loop shaped like that kernel's k-loop (barrier, operand
// conversion + LDS stores, barrier, LDS fragment reads + MFMAs) and checks every packed result against the scalar
// product in registers.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize pk_probe.hip -o pk_probe && ./pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode bit 0: an operand-select form (FORM; else a pre-broadcast pair, no op_sel);  bit 1: MFMAs in the loop;
// bit 2: LDS traffic;  bit 3: roles split by workgroup parity (even: packed ops + checks only, odd: MFMAs only)
// FORM 1: op_sel:[0,1] (low lane <- high dword of src1)   2: op_sel:[1,0] (low lane <- high dword of src0)
//      3: op_sel_hi:[1,0] (high lane <- low dword of src1) 4: op_sel_hi:[0,1] (high lane <- low dword of src0)
template <int MODE, int FORM>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, int n_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sc = smem;                       // [2] the two "scales", read back from LDS like the GEMM does
  _Float16* tile = reinterpret_cast<_Float16*>(smem + 4);
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { sc[0] = 2048.f; sc[1] = 524288.f; }
  __syncthreads();
  unsigned n_bad_lo = 0, n_bad_hi = 0, n_zero_lo = 0;
  f32x16 acc = {};
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool do_pk = !(MODE & 8) || (blockIdx.x & 1) == 0, do_mfma = !(MODE & 8) || (blockIdx.x & 1) == 1;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    const f32x2 s = *reinterpret_cast<const f32x2*>(sc);
    const f32x2 sb = {s.y, s.y};
#pragma unroll
    for (int e = 0; e < (do_pk ? 8 : 0); ++e) {
      const float4 r = *reinterpret_cast<const float4*>(in + (base + (size_t)(it * 8 + e) * 1024 * 1024) % n_in);
      const f32x2 a = {r.z, r.w};
      f32x2 p, q;
      const f32x2 c = {1.f, 2.f};
      const f32x2 t = {s.y, s.x};                      // for the forms that take the LOW dword
      if ((MODE & 1) && FORM == 1) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(a), "v"(s));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(s), "v"(c));
      } else if ((MODE & 1) && FORM == 2) {
        asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=v"(p) : "v"(a), "v"(s));
        asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(s), "v"(c));
      } else if ((MODE & 1) && FORM == 3) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(t));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(t), "v"(c));
      } else if ((MODE & 1) && FORM == 4) {
        asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(t), "v"(c));
      } else {
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(sb));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(sb), "v"(c));
      }
      const float wl = r.z * s.y, wh = r.w * s.y;
      n_bad_lo += (p.x != wl) + (q.x != wl - 1.f);
      n_bad_hi += (p.y != wh) + (q.y != wh - 2.f);
      n_zero_lo += (p.x == 0.f && wl != 0.f) + (q.x == -1.f && wl != 0.f);
      if (MODE & 4) {
        typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
        const h2_t h01 = __builtin_amdgcn_cvt_pkrtz(r.x * s.y, r.y * s.y), h23 = __builtin_amdgcn_cvt_pkrtz(p.x, p.y);
        uint2 w;
        w.x = __builtin_bit_cast(unsigned, h01); w.y = __builtin_bit_cast(unsigned, h23);
        *reinterpret_cast<uint2*>(tile + ((threadIdx.x >> 4) + 16 * e) * 72 + (threadIdx.x & 15) * 4) = w;
        *reinterpret_cast<uint2*>(tile + 128 * 72 + ((threadIdx.x >> 4) + 16 * e) * 72 + (threadIdx.x & 15) * 4) = w;
      }
    }
    __syncthreads();
    if ((MODE & 2) && do_mfma) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 fa, fb;
        if (MODE & 4) {
          fa = *reinterpret_cast<const f16x8*>(tile + (lane & 31) * 72 + (lane >> 5) * 8 + ks * 16);
          fb = *reinterpret_cast<const f16x8*>(tile + 128 * 72 + (lane & 31) * 72 + (lane >> 5) * 8 + ks * 16);
        } else {
          for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(float)(lane + j); fb[j] = (_Float16)(float)(it + j); }
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
      }
    }
  }
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += acc[r];
  if (sink == 12345.678f) n_bad_lo += 1;            // keeps the MFMAs alive
  atomicAdd(bad + 0 * 64 + lane, n_bad_lo);
  atomicAdd(bad + 1 * 64 + lane, n_bad_hi);
  atomicAdd(bad + 2 * 64 + lane, n_zero_lo);
}

template <int MODE, int FORM>
static void run(const float* d_in, int n_in, unsigned* d_bad, int blocks, int iters, size_t lds) {
  CHECK(hipMemset(d_bad, 0, 3 * 64 * sizeof(unsigned)));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipLaunchKernelGGL((probe<MODE, FORM>), dim3(blocks), dim3(256), lds, 0, d_in, d_bad, iters, n_in);
  CHECK(hipDeviceSynchronize());
  unsigned h[3 * 64];
  CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long q[3][4] = {};
  for (int k = 0; k < 3; ++k) for (int l = 0; l < 64; ++l) q[k][l / 16] += h[k * 64 + l];
  printf("form %d mode %d (%s%s%s%s) blocks %d lds %zu: low-lane mismatches by quarter-wave %llu %llu %llu %llu | high-lane %llu %llu %llu %llu | low-lane zero %llu %llu %llu %llu\n",
         FORM, MODE, (MODE & 1) ? "op_sel" : "broadcast", (MODE & 2) ? "+mfma" : "", (MODE & 4) ? "+lds" : "",
         (MODE & 8) ? ", roles split by workgroup parity" : "", blocks, lds,
         q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3], q[2][0], q[2][1], q[2][2], q[2][3]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;
  const int n_in = 64 << 20;
  float* h_in = (float*)malloc((size_t)n_in * 4);
  unsigned x = 12345;
  for (int i = 0; i < n_in; ++i) { x = x * 1664525u + 1013904223u; h_in[i] = ((int)(x >> 8) - (1 << 23)) * (1.f / (1 << 23)) * 0.0221f; }
  float* d_in; unsigned* d_bad;
  CHECK(hipMalloc(&d_in, (size_t)n_in * 4));
  CHECK(hipMalloc(&d_bad, 3 * 64 * sizeof(unsigned)));
  CHECK(hipMemcpy(d_in, h_in, (size_t)n_in * 4, hipMemcpyHostToDevice));
  const size_t lds2 = 77872, lds1 = 100 * 1024;      // two workgroups per CU (the GEMM's footprint) / one
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 1>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<3, 1>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<5, 1>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<7, 1>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<7, 1>(d_in, n_in, d_bad, 2048, iters, lds1);
    run<6, 0>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<3, 2>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<3, 3>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<3, 4>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<11, 1>(d_in, n_in, d_bad, 2048, iters, lds2);
    run<11, 3>(d_in, n_in, d_bad, 2048, iters, lds2);
  }
  return 0;
}
