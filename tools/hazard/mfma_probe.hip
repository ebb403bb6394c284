// What the matrix pipe of this part sustains, by itself and with the GEMM's other traffic beside it.  Not product code:
//   hipcc -O3 --offload-arch=gfx950 tools/hazard/mfma_probe.hip -o tools/_hazard/mfma_probe && tools/_hazard/mfma_probe
// One block per CU-slot (grid = 256 * blocks_per_cu... here 1 block of 8 waves per CU, x4 rounds), every wave runs
// ITER iterations of 48 v_mfma_f32_32x32x16_f16 on 8 accumulator tiles (the layer-0 GEMM's per-k-tile MFMA work), plus,
// by mode bit:  1: 24 ds_read_b128 per iteration (the fragment reads, conflict-free addresses, results feed the MFMAs)
//               2: two s_barrier per iteration
//               4: 8 global_load_lds_dwordx4 per iteration (the tile DMA; streams a 400 MB buffer)
//               8: 4 waves per block (one wave per SIMD) instead of 8
//              16: MFMAs replaced by nothing (the other traffic alone)
//              64: LDS filled with random signs and exponents (not just random mantissas in [0.5, 1))
//              32: ds_reads in the GEMM's fragment pattern (64-byte rows, XOR-swizzled slots) instead of linear
// Prints the fp16 MFMA rate per mode.  peak = 256 CU x 4 SIMD x 1024 flop/clk x clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(const _Float16* __restrict__ src, float* __restrict__ out, int iters,
                                             long src_elems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 fa[2][4][2], fb[2][2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[ks][i][q][e] = (_Float16)(0.001f * (lane + i + q));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[ks][j][q][e] = (_Float16)(0.002f * (lane + j + q));
  }
  if (MODE & 1) {                                              // defined LDS contents
    for (int o = threadIdx.x * 16; o < 131072; o += blockDim.x * 16) {
      unsigned h = (unsigned)o * 2654435761u;                  // halves in [0.5, 1): exponent 0x38, random mantissas
      uint4 v;
      v.x = 0x38003800u | (h & 0x03ff03ffu);
      v.y = 0x38003800u | ((h >> 3) & 0x03ff03ffu);
      v.z = 0x38003800u | ((h >> 5) & 0x03ff03ffu);
      v.w = 0x38003800u | ((h >> 7) & 0x03ff03ffu);
      if (MODE & 64) {                                         // random signs, exponents 2^-8 .. 2^0, random mantissas
        auto wide = [](unsigned x) {
          x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
          const unsigned lo = (x & 0x83ffu) | ((0x1c + ((x >> 10) & 7) + ((x >> 13) & 1)) << 10);
          const unsigned y = x >> 16;
          const unsigned hi = (y & 0x83ffu) | ((0x1c + ((y >> 10) & 7) + ((y >> 13) & 1)) << 10);
          return lo | (hi << 16);
        };
        v.x = wide(h); v.y = wide(h + 1); v.z = wide(h + 2); v.w = wide(h + 3);
      }
      *(uint4*)(smem + o) = v;
    }
    __syncthreads();
  }
  // 64 lanes x 16 B contiguous, or (bit 32) the GEMM's fragment pattern: lane -> row (lane & 31) of 64-byte rows, 16-byte
  // slot (lane >> 5) ^ ((row >> 2) & 3)
  const unsigned lds_lane = (MODE & 32) ? (unsigned)(size_t)(smem) + (wid & 1) * 8192 + (lane & 31) * 64 + (((lane >> 5) ^ ((lane >> 2) & 3)) * 16)
                                        : (unsigned)(size_t)(smem) + wid * 4096 + lane * 16;
  const long stride = (long)gridDim.x * blockDim.x * 8;        // halves per DMA wavefront sweep
  long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE & 4) {
      unsigned char* st = smem + 65536 + wid * 1024;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + g),
                                         (__attribute__((address_space(3))) void*)(st + j * 8192), 16, 0, 0);
        g += stride;
        if (g + 8 > src_elems) g -= (src_elems / stride) * stride;
      }
    }
    if (MODE & 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[ks][i][q]) : "v"(lds_lane), "n"((ks * 12 + i * 2 + q) * 1024));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[ks][j][q]) : "v"(lds_lane), "n"((ks * 12 + 8 + j * 2 + q) * 1024));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE & 2) __builtin_amdgcn_s_barrier();
    if (MODE & 128) {                                           // the same flops as 96 v_mfma_f32_16x16x32_f16 (16 passes... 8 clocks each x2)
      f32x4* a4 = reinterpret_cast<f32x4*>(acc);                 // 32 accumulator tiles of 16x16
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              a4[(i * 2 + j) * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][i][pr == 2], fb[ks][j][pr == 1], a4[(i * 2 + j) * 4 + 0], 0, 0, 0);
              a4[(i * 2 + j) * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][i][pr != 2], fb[ks][j][pr == 1], a4[(i * 2 + j) * 4 + 1], 0, 0, 0);
            }
    } else if (!(MODE & 16)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][pr == 2], fb[ks][j][pr == 1], acc[i * 2 + j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[0][i][0]), "+v"(fa[1][i][1]));
    }
    if (MODE & 4) __builtin_amdgcn_s_waitcnt(0x0F70);
    if (MODE & 2) __builtin_amdgcn_s_barrier();
  }
  const uint64_t t_end = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (float)(t_end - t_begin);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (MODE & 16) s += (float)fa[0][0][0][0] + (float)fa[1][3][1][0] + (float)fb[1][1][1][0];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const _Float16* src, float* out, long src_elems, int iters) {
  const int threads = (MODE & 8) ? 256 : 512;
  const int grid = 256 * 4;
  CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  probe<MODE><<<grid, threads, 131072>>>(src, out, iters, src_elems);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a));
    probe<MODE><<<grid, threads, 131072>>>(src, out, iters, src_elems);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double waves = (double)grid * threads / 64;
  const double flop = waves * iters * 48.0 * 32768.0;
  const double dma = (MODE & 4) ? waves * iters * 8.0 * 1024.0 : 0.0;
  printf("mode %2d (%s%s%s%s%s%s): %8.3f ms", MODE, (MODE & 16) ? "no-mfma " : ((MODE & 128) ? "mfma16x16x32 " : "mfma "), (MODE & 1) ? ((MODE & 32) ? "ds_read(gemm pattern) " : "ds_read ") : "",
         (MODE & 2) ? "barrier " : "", (MODE & 4) ? "dma " : "", (MODE & 64) ? "wide-data " : "", (MODE & 8) ? "1wave/simd" : "2waves/simd", best);
  if (!(MODE & 16)) printf("  %7.1f TFLOP/s fp16", flop / best * 1e-9);
  if (dma > 0) printf("  DMA %6.2f TB/s", dma / best * 1e-9);
  float counts;
  CK(hipMemcpy(&counts, out + 1, 4, hipMemcpyDeviceToHost));
  // block 0 is one of grid/256 blocks a CU runs back to back: its loop takes about best / (grid / 256)
  printf("  | s_memtime: %.0f counts in block 0's loop = %.2f counts/ns", counts, counts / (best * 1e6 / (grid / 256.0)));
  if (!(MODE & 16)) printf(", %.1f per MFMA of its SIMD", counts / (iters * 48.0 * ((MODE & 8) ? 1 : 2)));
  printf("\n");
}

int main() {
  const long elems = 200L << 20;                               // 400 MB of halves
  _Float16* src;
  float* out;
  CK(hipMalloc(&src, elems * 2));
  CK(hipMemset(src, 0, elems * 2));
  CK(hipMalloc(&out, 64));
  const int iters = 512;
  run<0>(src, out, elems, iters);
  run<8>(src, out, elems, iters);
  run<1>(src, out, elems, iters);
  run<3>(src, out, elems, iters);
  run<4>(src, out, elems, iters);
  run<5>(src, out, elems, iters);
  run<7>(src, out, elems, iters);
  run<17>(src, out, elems, iters);
  run<49>(src, out, elems, iters);
  run<33>(src, out, elems, iters);
  run<35>(src, out, elems, iters);
  run<99>(src, out, elems, iters);
  run<99 + 128>(src, out, elems, iters);
  run<35 + 128>(src, out, elems, iters);
  run<20>(src, out, elems, iters);
  run<23>(src, out, elems, iters);
  return 0;
}
