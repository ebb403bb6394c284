// Stand-alone scatter_{add,mean,max}(src[E,C], index[E], dim=0, dim_size) for gfx950: the functional
// surface of the third-party op the reference aggregates with (pytorch-scatter 2.0.8; call sites
// reference models/mpn.py:196,199,202).  Inside MOTMPNet.forward the aggregation is fused into
// pass_c_kernel; these kernels serve callers that use the op on its own.
//   add : out zero-filled, out[index[e], c] += src[e, c]
//   mean: add, then divide by max(count, 1)
//   max : per-column maximum, rows nobody writes stay 0, argmax = smallest e attaining it (E if none)
#include "kernels.h"
#include <limits.h>

namespace mtmc {

__device__ __forceinline__ int float_key(float f) {       // order-preserving float -> signed int
  const int b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff)); }

__global__ void scatter_fill_kernel(float* out, int64_t n, int bits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = __int_as_float(bits);
}
__global__ void scatter_fill_i64_kernel(int64_t* out, int64_t n, int64_t v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

__global__ void scatter_add_kernel(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                                   int64_t dim_size, float* out, float* count) {
  const int64_t total = n_src * n_cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / n_cols, c = i % n_cols;
    const int64_t r = index[e];
    if (r < 0 || r >= dim_size) continue;
    unsafeAtomicAdd(out + r * n_cols + c, src[i]);
    if (count && c == 0) unsafeAtomicAdd(count + r, 1.0f);
  }
}
__global__ void scatter_div_kernel(float* out, const float* count, int64_t dim_size, int64_t n_cols) {
  const int64_t total = dim_size * n_cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float cnt = count[i / n_cols];
    out[i] = out[i] / (cnt < 1.f ? 1.f : cnt);
  }
}
__global__ void scatter_max_kernel(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                                   int64_t dim_size, float* out) {
  const int64_t total = n_src * n_cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / n_cols, c = i % n_cols;
    const int64_t r = index[e];
    if (r < 0 || r >= dim_size) continue;
    atomicMax(reinterpret_cast<int*>(out) + r * n_cols + c, float_key(src[i]));
  }
}
__global__ void scatter_argmax_kernel(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                                      int64_t dim_size, const float* out_keys, int64_t* arg) {
  const int64_t total = n_src * n_cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t e = i / n_cols, c = i % n_cols;
    const int64_t r = index[e];
    if (r < 0 || r >= dim_size) continue;
    if (float_key(src[i]) == reinterpret_cast<const int*>(out_keys)[r * n_cols + c])
      atomicMin(reinterpret_cast<unsigned long long*>(arg) + r * n_cols + c, (unsigned long long)e);
  }
}
__global__ void scatter_unkey_kernel(float* out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int k = reinterpret_cast<int*>(out)[i];
    out[i] = (k == INT_MIN) ? 0.f : key_float(k);
  }
}

// the 1-D integer call form of the post-processing (reference utils.py:173-174, :308-309): int64 in, int64 out, exact
__global__ void scatter_add_i64_kernel(const int64_t* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                                       int64_t dim_size, int64_t* out) {
  const int64_t total = n_src * n_cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = index[i / n_cols];
    if (r < 0 || r >= dim_size) continue;
    atomicAdd(reinterpret_cast<unsigned long long*>(out) + r * n_cols + i % n_cols, (unsigned long long)src[i]);
  }
}

static inline int sgrid(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// mode 0 add, 1 mean, 2 max
void launch_scatter(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                    float* out, float* count, int64_t* arg_out, int mode, hipStream_t s) {
  const int64_t n_out = dim_size * n_cols, n_in = n_src * n_cols;
  if (mode == 2) {
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(sgrid(n_out)), dim3(256), 0, s, out, n_out, INT_MIN);
    if (n_in > 0)
      hipLaunchKernelGGL(scatter_max_kernel, dim3(sgrid(n_in)), dim3(256), 0, s, src, index, n_src, n_cols, dim_size, out);
    if (arg_out) {
      hipLaunchKernelGGL(scatter_fill_i64_kernel, dim3(sgrid(n_out)), dim3(256), 0, s, arg_out, n_out, n_src);
      if (n_in > 0)
        hipLaunchKernelGGL(scatter_argmax_kernel, dim3(sgrid(n_in)), dim3(256), 0, s, src, index, n_src, n_cols,
                           dim_size, out, arg_out);
    }
    hipLaunchKernelGGL(scatter_unkey_kernel, dim3(sgrid(n_out)), dim3(256), 0, s, out, n_out);
    return;
  }
  hipLaunchKernelGGL(scatter_fill_kernel, dim3(sgrid(n_out)), dim3(256), 0, s, out, n_out, 0);
  if (mode == 1) hipLaunchKernelGGL(scatter_fill_kernel, dim3(sgrid(dim_size)), dim3(256), 0, s, count, dim_size, 0);
  if (n_in > 0)
    hipLaunchKernelGGL(scatter_add_kernel, dim3(sgrid(n_in)), dim3(256), 0, s, src, index, n_src, n_cols, dim_size, out,
                       mode == 1 ? count : nullptr);
  if (mode == 1)
    hipLaunchKernelGGL(scatter_div_kernel, dim3(sgrid(n_out)), dim3(256), 0, s, out, count, dim_size, n_cols);
}

void launch_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                            int64_t* out, hipStream_t s) {
  hipLaunchKernelGGL(scatter_fill_i64_kernel, dim3(sgrid(dim_size * n_cols)), dim3(256), 0, s, out, dim_size * n_cols, (int64_t)0);
  if (n_src * n_cols > 0)
    hipLaunchKernelGGL(scatter_add_i64_kernel, dim3(sgrid(n_src * n_cols)), dim3(256), 0, s, src, index, n_src, n_cols,
                       dim_size, out);
}

}  // namespace mtmc
