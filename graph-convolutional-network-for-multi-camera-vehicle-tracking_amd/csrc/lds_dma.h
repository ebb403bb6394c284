// Shared by the pre-split GEMM kernels (gemm_presplit.hip, gemm_staged.hip, lab/presplit_lab.hip): the plane layout
// constant, the vector types and the LDS-DMA instruction wrapper.
#pragma once
#include <hip/hip_runtime.h>

namespace mtmc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __fp16 h2_t __attribute__((ext_vector_type(2)));

// The planes are stored K-TILE-MAJOR and PRE-SWIZZLED: the eight halves k .. k+7 (k % 8 == 0) of row `row` of a plane
// of `rows` rows live at
//     ((k / kPlaneKT) * rows + row) * kPlaneKT + 8 * (((k % kPlaneKT) / 8) ^ plane_swz(row)),
// i.e. memory holds the LDS image itself (bank swizzle included; tile origins are multiples of 16 rows), so the
// [256 rows][32 halves] image a GEMM block stages per k-tile is ONE contiguous 16 KB run that LDS-DMA copies in lane
// order: every global_load_lds_dwordx4 reads 1 KB of consecutive addresses, whole 128-byte lines.  Row-major planes made each such instruction touch
// sixteen 64-byte half lines 2*K bytes apart and the GEMM ran at the rate the CUs' L1s could be fed in half lines:
// 1.28 ms against 0.94 ms at 100000 x 2048 x 1024.
constexpr int kPlaneKT = 32;

// The bank swizzle of a [rows][32 halves] image (64-byte rows: four rows per 256-byte bank row): the 16-byte slot s of row r
// is stored at slot s ^ plane_swz(r).  Rounds 2-4 used (r >> 2) & 3, chosen for the 32 x 32 x 16 fragment read (lane l: row
// l % 32, slot 2 ks + l / 32) -- and kept it when the product kernels moved to 16 x 16 x 32 MFMAs, whose fragment read (lane l:
// row l % 16, slot l / 16) is 2-WAY conflicted on it: ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ... (MI355X_MICROARCH.md, LDS), and rows r and r + 4 with slots s and s ^ 1 share their four banks.  Round 5: the
// permutation {0, 2, 3, 1} of (r >> 2) & 3 is conflict-free for BOTH read shapes (exhaustive search over the 4^8 functions of
// row bits 2-4 against the guide's lane groups: tools/plane_swizzle_search.py).  MTMC_PLANE_SWZ_OLD=1 builds the old image (A/B).
#ifndef MTMC_PLANE_SWZ_OLD
#define MTMC_PLANE_SWZ_OLD 0
#endif
__host__ __device__ __forceinline__ constexpr int plane_swz(int row) {
  return MTMC_PLANE_SWZ_OLD ? ((row >> 2) & 3) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
}

// One LDS-DMA instruction (64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024) in lane order) in the form that costs the
// issuing wave NO VALU instruction: uniform 64-bit base in SGPRs + a per-lane 32-bit byte offset that is loop-invariant.
// hipcc selects the 64-bit-VGPR-address form for __builtin_amdgcn_global_load_lds here (one v_lshl_add_u64 per
// instruction).  Two things the compiler does NOT know about this statement, both checked on the built code object by
// tools/check_isa.py (rule LDS-DMA-M0): (i) it writes m0 -- hipcc treats m0 as reserved and sets it itself right before each
// use of its own, so nothing may sit between this s_mov_b32 m0 and its global_load_lds that reads or writes m0; (ii) it is a
// vector-memory operation the compiler does not count in vmcnt: every wait on it is written out by hand.
__device__ __forceinline__ void lds_dma16(const void* base, unsigned lane_off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds) : "memory");
}

}  // namespace mtmc
