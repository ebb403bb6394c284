// Persistent round loop vs kernel boundaries, measured in isolation ("one persistent
// kernel (or a captured HIP graph) is required for a meaningful edges/s" on few-edge graphs).
//
// The S02 forward is 25 dependent launches; 15 of them are the three message-passing rounds (node_proj, pass A, pass B,
// node_stat, pass C), each a grid-wide dependency: every kernel ends by adding its BatchNorm statistics into replicated fp64
// blocks and the next one begins by gathering them.  A persistent launch would replace the 15 kernel boundaries by 15 grid
// barriers.  This probe runs exactly that skeleton both ways on the same box, with the statistics hand-off of the real
// kernels (74 doubles into 16 replicas at the end of a phase, gathered by every workgroup at the start of the next) and NO
// other work, so the difference between the two columns is what the synchronisation structure itself costs or saves:
//
//   launches     15 trivial kernels back to back (each: gather, `spin` ns of stand-in work, block-reduced atomics)
//   flat         one launch, 15 x a single-counter grid barrier (release fence before the arrive, acquire fence after)
//   xcd          one launch, 15 x the XCD-hierarchical barrier of MI355X_MICROARCH.md ("barrier-xcd": per-XCC arrival
//                counter, the last arriver of an XCC releases and arrives at the top counter, waits for all XCCs, then
//                opens its XCC's generation word; every workgroup acquires) -- never cooperative_groups grid sync
//
//   hipcc -O2 --offload-arch=gfx950 tools/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe [blocks_per_cu] [spin_ns]
//
// Every spin is bounded (a barrier that does not complete within ~2 ms sets a flag and returns), the grid is sized from the
// occupancy API minus a margin, and each phase checks a value another workgroup wrote before the barrier.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }    \
  } while (0)

constexpr int kRep = 16, kStat = 74, kStride = 80;   // replicas x doubles per statistics block (a round's largest: 10 + 64)
constexpr int kPhases = 15;
constexpr int kLine = 32;                            // unsigned per 128-byte line: every counter on a line of its own

struct Bar {
  unsigned xcc_arrive[8 * kLine];
  unsigned xcc_size[8 * kLine];
  unsigned gen[8 * kLine];
  unsigned top[kLine];
  unsigned flat[kLine];
  unsigned census[kLine];
  unsigned timeout[kLine];
  unsigned stale[kLine];
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, unsigned* timeout) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  while (ld_agent(p) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {            // 2 ms: give up, report
      __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

// every thread: own stores done -> workgroup barrier -> lane 0 runs the protocol -> workgroup barrier
__device__ void flat_barrier(Bar* b, unsigned n_blocks, unsigned phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    add_agent(&b->flat[0], 1u);
    spin_until(&b->flat[0], phase * n_blocks, &b->timeout[0]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ void xcd_barrier(Bar* b, int xcc, unsigned xsize, unsigned n_xcc, unsigned phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = add_agent(&b->xcc_arrive[xcc * kLine], 1u);
    if (prev == phase * xsize - 1u) {                         // last arriver of this XCC: its L2 holds the XCC's stores
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      add_agent(&b->top[0], 1u);
      spin_until(&b->top[0], phase * n_xcc, &b->timeout[0]);
      __hip_atomic_store(&b->gen[xcc * kLine], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(&b->gen[xcc * kLine], phase, &b->timeout[0]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// one phase's body: gather the previous phase's statistics (as stat_gather does), stand-in work, add this phase's
__device__ void phase_body(double* stats, int phase, int spin_ns, double* sink, const unsigned* mark, unsigned* stale,
                           unsigned* my_mark, int n_blocks) {
  __shared__ double st[kStat];
  if (phase > 0) {
    const double* prev = stats + (size_t)(phase - 1) * kRep * kStride;
    if (threadIdx.x < kStat) {
      double v = 0;
      for (int r = 0; r < kRep; ++r) v += prev[r * kStride + threadIdx.x];
      st[threadIdx.x] = v;
    }
    // visibility check: the workgroup "opposite" in the grid wrote phase into its mark before the barrier
    if (threadIdx.x == 0 && mark[((blockIdx.x + n_blocks / 2 + 3) % n_blocks) * kLine] != (unsigned)phase) atomicAdd(stale, 1u);
  }
  __syncthreads();
  if (spin_ns > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((__builtin_amdgcn_s_memrealtime() - t0) * 10ull < (unsigned long long)spin_ns) __builtin_amdgcn_s_sleep(1);
  }
  double* mine = stats + (size_t)phase * kRep * kStride + (blockIdx.x % kRep) * kStride;
  if (threadIdx.x < kStat) unsafeAtomicAdd(mine + threadIdx.x, 1.0 + (phase > 0 ? st[threadIdx.x] * 1e-9 : 0.0));
  if (threadIdx.x == 0) {
    my_mark[blockIdx.x * kLine] = (unsigned)(phase + 1);
    if (phase == kPhases - 1) sink[blockIdx.x] = st[0];
  }
}

__global__ __launch_bounds__(256) void one_phase_kernel(double* stats, int phase, int spin_ns, double* sink, unsigned* mark,
                                                        unsigned* stale) {
  phase_body(stats, phase, spin_ns, sink, mark, stale, mark, gridDim.x);
}

template <int MODE>   // 0 flat, 1 xcd
__global__ __launch_bounds__(256) void persistent_kernel(double* stats, int spin_ns, double* sink, unsigned* mark, Bar* bar) {
  int xcc = 0;
  unsigned xsize = 0, n_xcc = 0;
  if (MODE == 1) {
    // census: which XCC am I on, how many workgroups share it (placement is not guaranteed even), how many XCCs are in use
    __shared__ unsigned sh[3];
    if (threadIdx.x == 0) {
      xcc = (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7);     // HW_REG_XCC_ID, bits [3:0]
      add_agent(&bar->xcc_size[xcc * kLine], 1u);
      add_agent(&bar->census[0], 1u);
      spin_until(&bar->census[0], gridDim.x, &bar->timeout[0]);
      unsigned n = 0;
      for (int x = 0; x < 8; ++x) n += ld_agent(&bar->xcc_size[x * kLine]) > 0 ? 1u : 0u;
      sh[0] = (unsigned)xcc; sh[1] = ld_agent(&bar->xcc_size[xcc * kLine]); sh[2] = n;
    }
    __syncthreads();
    xcc = (int)sh[0]; xsize = sh[1]; n_xcc = sh[2];
  }
  for (int ph = 0; ph < kPhases; ++ph) {
    phase_body(stats, ph, spin_ns, sink, mark, &bar->stale[0], mark, gridDim.x);
    if (ph + 1 < kPhases) {
      if (MODE == 0) flat_barrier(bar, gridDim.x, (unsigned)(ph + 1));
      else xcd_barrier(bar, xcc, xsize, n_xcc, (unsigned)(ph + 1));
    }
  }
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const int spin_ns = argc > 2 ? atoi(argv[2]) : 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<1>, 256, 0));
  if (per_cu < 1 || per_cu > 4 || per_cu > occ - 1) { fprintf(stderr, "blocks per CU %d not safely resident (occupancy API %d)\n", per_cu, occ); return 1; }
  const int n_blocks = prop.multiProcessorCount * per_cu;
  double *stats, *sink;
  unsigned* mark;
  Bar* bar;
  const size_t stat_bytes = (size_t)kPhases * kRep * kStride * sizeof(double);
  CHECK(hipMalloc(&stats, stat_bytes));
  CHECK(hipMalloc(&sink, n_blocks * sizeof(double)));
  CHECK(hipMalloc(&mark, (size_t)n_blocks * kLine * sizeof(unsigned)));
  CHECK(hipMalloc(&bar, sizeof(Bar)));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto reset = [&]() {                                  // what the forward's one memset node does
    CHECK(hipMemsetAsync(stats, 0, stat_bytes, s));
    CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    CHECK(hipMemsetAsync(mark, 0, (size_t)n_blocks * kLine * sizeof(unsigned), s));
  };
  auto run = [&](int mode) {
    if (mode == 0) {
      for (int ph = 0; ph < kPhases; ++ph)
        hipLaunchKernelGGL(one_phase_kernel, dim3(n_blocks), dim3(256), 0, s, stats, ph, spin_ns, sink, mark, &bar->stale[0]);
    } else if (mode == 1) {
      hipLaunchKernelGGL(persistent_kernel<0>, dim3(n_blocks), dim3(256), 0, s, stats, spin_ns, sink, mark, bar);
    } else {
      hipLaunchKernelGGL(persistent_kernel<1>, dim3(n_blocks), dim3(256), 0, s, stats, spin_ns, sink, mark, bar);
    }
  };
  const char* names[3] = {"launches", "flat", "xcd"};
  printf("grid %d workgroups (%d per CU), %d phases, stand-in work %d ns per phase\n", n_blocks, per_cu, kPhases, spin_ns);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> t;
    for (int it = 0; it < 60; ++it) {
      reset();
      CHECK(hipEventRecord(e0, s));
      run(mode);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 10) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    Bar h;
    CHECK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    std::vector<double> hs(kRep * kStride);
    CHECK(hipMemcpy(hs.data(), stats + (size_t)(kPhases - 1) * kRep * kStride, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
    double tot = 0;
    for (int r = 0; r < kRep; ++r) tot += hs[r * kStride];
    printf("%-9s median %7.2f us  p10 %7.2f  p90 %7.2f   per phase %5.2f us   timeouts %u  stale reads %u  last-phase count %.0f (want %d)\n",
           names[mode], t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], t[t.size() / 2] / kPhases, h.timeout[0], h.stale[0],
           tot, n_blocks);
    if (h.timeout[0]) { printf("a barrier timed out: stopping\n"); return 2; }
  }
  return 0;
}
