// The edge part of prep_kernel as a device function: shared by prep_kernel (edge_kernels.hip) and by the passenger workgroups of
// few_l0_kernel (gemm_few.hip: on few-row forwards nothing of this is needed before the first round, so it rides in the first
// encoder layer's launch instead of standing in front of it).  NW = waves per workgroup of the hosting kernel.
#pragma once
#include "kernels.h"

namespace mtmc {

template <int NW>
__device__ __forceinline__ void prep_edge_body(const PrepEdge& p, int eb) {
  __shared__ double red[5 * NW];
  double acc[5] = {0, 0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const int64_t nthreads = (int64_t)p.n_edge_blocks * blockDim.x;
  // whole waves iterate together so that the run-length logic sees 64 consecutive edges
  const int64_t e_end = ((p.n_edges + 63) / 64) * 64;
  for (int64_t e = (int64_t)eb * blockDim.x + threadIdx.x; e < e_end; e += nthreads) {
    const bool active = e < p.n_edges;
    int64_t r64 = 0, c64 = 0;
    if (active) {
      r64 = p.row[e * p.idx_stride];
      c64 = p.col[e * p.idx_stride];
      if (r64 < 0 || r64 >= p.n_nodes || c64 < 0 || c64 >= p.n_nodes) {
        p.flags[1] = 1;                                   // out of range: clamp, report
        r64 = r64 < 0 ? 0 : (r64 >= p.n_nodes ? p.n_nodes - 1 : r64);
        c64 = c64 < 0 ? 0 : (c64 >= p.n_nodes ? p.n_nodes - 1 : c64);
      }
      p.row32[e] = (int)r64;
      p.col32[e] = (int)c64;
      float a0, a1;
      load_attr(p.attr, p.fe, e, a0, a1);
      acc[0] += a0; acc[1] += a1;
      acc[2] += (double)a0 * a0; acc[3] += (double)a0 * a1; acc[4] += (double)a1 * a1;
    }
    const int r = active ? (int)r64 : -1;
    int prev = __shfl_up(r, 1, 64);
    int64_t prev_c = __shfl_up(c64, 1, 64);
    if (lane == 0) {
      prev = (active && e > 0) ? (int)p.row[(e - 1) * p.idx_stride] : r;
      prev_c = (active && e > 0) ? p.col[(e - 1) * p.idx_stride] : c64;
    }
    if (active && prev > r) p.flags[0] = 1;               // rows not globally non-decreasing
    if (active && prev == r && prev_c > c64) p.flags[2] = 1;   // columns not ascending inside a row (column-blocked pass A)
    if (active && (e == 0 || prev != r)) p.row_start[r] = (int)e;   // first edge of the row (meaningful when sorted)
    // out-degree: one atomic per run of equal rows inside the wave
    const bool head = active && (lane == 0 || prev != r);
    const unsigned long long heads = __ballot(head);
    const int n_active = __popcll(__ballot(active));
    if (head) {
      const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
      const int len = above ? (__ffsll((long long)above)) : (n_active - lane);
      atomicAdd(p.deg + r, len);
    }
  }
  block_atomic_add<5>(acc, p.stat_attr, kAttrStride, red, eb);
}

}  // namespace mtmc
