// relu(bn(Y)) (+ Dropout) of 16 rows x 64 columns per workgroup with a TRANSPOSED copy beside it: the body of
// bn_relu_rows_t_kernel (node_kernels.hip) and of the z = 1 workgroups of bn_bwd_kernel<1> (train_kernels.hip: the backward's
// recomputation of a layer's input activation rides in the launch that applies the layer's BatchNorm backward -- the two
// are independent).  A thread keeps ONE column: its BatchNorm affine is derived once; the transposed tile leaves through LDS
// as one 16-byte store per thread.  bx / by / ny: the workgroup's tile and the number of row tiles.
#pragma once
#include "common.h"

namespace mtmc {

struct RowsTJob {
  const float* Y; int64_t ldy; int64_t rows; int dim;
  const double* stats; const float* gamma; const float* beta; double count;
  float* dst; Drop drop; unsigned drop_stream; int64_t row0; unsigned* amax_out; float* dstT; int64_t ldt;
};

__device__ __forceinline__ void bn_relu_rows_t_body(const RowsTJob& j, int bx, int by, int ny) {
  __shared__ float tile[16][65];
  __shared__ float wmax[4];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = bx * 64 + cl;
  const int64_t r0 = (int64_t)by * 16;
  float vmax = 0.f;
  float s = 0.f, t = 0.f;
  if (col < j.dim) bn_affine(j.stats[col], j.stats[j.dim + col], j.count, j.gamma[col], j.beta[col], s, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + rg + 4 * i;
    float v = 0.f;
    if (r < j.rows && col < j.dim) {
      v = drop_apply(j.drop, j.drop_stream, (unsigned long long)(j.row0 + r) * j.dim + col,
                     fmaxf(fmaf(j.Y[r * j.ldy + col], s, t), 0.f));
      j.dst[r * j.dim + col] = v;
    }
    tile[rg + 4 * i][cl] = v;                       // (rows past the end: the transposed copy's zero padding)
    vmax = fmaxf(vmax, v);
  }
  __syncthreads();
  {
    const int c = threadIdx.x >> 2, q = threadIdx.x & 3;      // column of the tile, quarter of its 16 rows
    if (bx * 64 + c < j.dim && r0 + 4 * q < j.ldt)            // (ldt % 4 == 0: a quarter is inside or outside as a whole)
      *reinterpret_cast<float4*>(j.dstT + (int64_t)(bx * 64 + c) * j.ldt + r0 + 4 * q) =
          make_float4(tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]);
  }
  if (by == ny - 1 && col < j.dim)                          // padding rows behind the last tile
    for (int64_t r = r0 + 16 + rg; r < j.ldt; r += 4) j.dstT[(int64_t)col * j.ldt + r] = 0.f;
  if (j.amax_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if (cl == 0) wmax[rg] = vmax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_publish(j.amax_out + (bx + by) % kAmaxRep, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

}  // namespace mtmc
