// Edge-classification loss of the training callers: F.cross_entropy / nn.CrossEntropyLoss(weight, reduction) on the
// [E, C<=4] logits the MPN emits (reference train.py:88-93, :109-142, :178-186).  torch's nll_loss kernels reduce with a
// single block (126 us forward + 54 us backward per classified step on the 173k-edge training graph, i.e. more than
// the whole MPN forward); here log-softmax, weighting and reduction are one pass forward and one pass backward.
//   forward : l_i = w[y_i] * (logsumexp(x_i) - x_i[y_i]);  sums = (sum_i l_i, sum_i w[y_i])  in fp64
//   backward: d x_i[c] = g_i * w[y_i] * (softmax(x_i)[c] - [c == y_i]),
//             g_i = grad / sum_w (mean), grad (sum), grad_i (none);  rows with y_i == ignore_index contribute nothing
#include "kernels.h"
#include "../../include/mtmc_mpn.h"

namespace mtmc {

__device__ __forceinline__ void ce_row(const float* x, int C, float (&p)[MTMC_MAX_CLASSES], float& lse) {
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) { p[c] = expf(x[c] - m); s += p[c]; }
  lse = m + logf(s);
  const float inv = 1.f / s;
  for (int c = 0; c < C; ++c) p[c] *= inv;
}

__global__ __launch_bounds__(256) void ce_forward_kernel(const float* logits, const int64_t* labels, const float* weight,
                                                         int64_t n, int C, int64_t ignore_index, float* per_sample,
                                                         double* sums, int64_t period) {
  __shared__ double red[2 * 4];
  double acc[2] = {0, 0};
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    const int64_t y = labels[period ? i % period : i];    // period: the same labels for every classified step
    float l = 0.f, w = 0.f;
    if (y != ignore_index && y >= 0 && y < C) {
      float x[MTMC_MAX_CLASSES], p[MTMC_MAX_CLASSES], lse;
      if (C == 2) { const float2 v = reinterpret_cast<const float2*>(logits)[i]; x[0] = v.x; x[1] = v.y; }
      else for (int c = 0; c < C; ++c) x[c] = logits[i * C + c];
      ce_row(x, C, p, lse);
      w = weight ? weight[y] : 1.f;
      l = w * (lse - x[y]);
    }
    if (per_sample) per_sample[i] = l;
    acc[0] += l;
    acc[1] += w;
  }
  block_atomic_add<2>(acc, sums, 2, red);
}

// totals to sums[0..1]; loss = mean or sum (one thread: 32 additions)
__global__ void ce_finalize_kernel(double* sums, int mode, float* loss_out, double scale) {
  double s1 = 0, s2 = 0;
  for (int r = 0; r < kStatRep; ++r) { s1 += sums[r * 2]; s2 += sums[r * 2 + 1]; }
  sums[0] = s1; sums[1] = s2;
  if (loss_out) loss_out[0] = (float)(mode == 0 ? scale * s1 / s2 : s1);
}

// mode: 0 = mean, 1 = sum, 2 = none (grad is [n] then)
__global__ __launch_bounds__(256) void ce_backward_kernel(const float* logits, const int64_t* labels, const float* weight,
                                                          int64_t n, int C, int64_t ignore_index, int mode,
                                                          const float* grad, const double* sums, float* d_logits,
                                                          int64_t period, double scale) {
  const float g_all = mode == 0 ? (float)(scale * (double)grad[0] / sums[1]) : (mode == 1 ? grad[0] : 0.f);
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    const int64_t y = labels[period ? i % period : i];
    float d[MTMC_MAX_CLASSES] = {0.f, 0.f, 0.f, 0.f};
    if (y != ignore_index && y >= 0 && y < C) {
      float x[MTMC_MAX_CLASSES], p[MTMC_MAX_CLASSES], lse;
      if (C == 2) { const float2 v = reinterpret_cast<const float2*>(logits)[i]; x[0] = v.x; x[1] = v.y; }
      else for (int c = 0; c < C; ++c) x[c] = logits[i * C + c];
      ce_row(x, C, p, lse);
      const float g = (mode == 2 ? grad[i] : g_all) * (weight ? weight[y] : 1.f);
      for (int c = 0; c < C; ++c) d[c] = g * (p[c] - (c == y ? 1.f : 0.f));
    }
    if (C == 2) reinterpret_cast<float2*>(d_logits)[i] = make_float2(d[0], d[1]);
    else for (int c = 0; c < C; ++c) d_logits[i * C + c] = d[c];
  }
}

// Confusion counts of argmax(logits) against 0/1 labels in one pass: what the training / validation loops derive their
// FPR, TPR, precision and recall from (reference train.py:98-107, inference.py:20-67) with boolean-mask indexing, i.e.
// with a device synchronisation per mask.  counts = {TP, FP, TN, FN}; labels other than 0 / 1 are skipped.
__global__ __launch_bounds__(256) void confusion_kernel(const float* logits, const int64_t* labels, int64_t n, int C,
                                                        unsigned long long* counts) {
  __shared__ unsigned int sh[4];
  if (threadIdx.x < 4) sh[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[4] = {0, 0, 0, 0};
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
    const int64_t y = labels[i];
    if (y != 0 && y != 1) continue;
    int best = 0;
    float bv = logits[i * C];
    for (int k = 1; k < C; ++k) { const float v = logits[i * C + k]; if (v > bv) { bv = v; best = k; } }   // first maximum
    const int pred = best == 1;
    ++c[y == 1 ? (pred ? 0 : 3) : (pred ? 1 : 2)];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned int v = c[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh[j], v);
  }
  __syncthreads();
  if (threadIdx.x < 4 && sh[threadIdx.x]) atomicAdd(counts + threadIdx.x, (unsigned long long)sh[threadIdx.x]);
}

static inline int ce_grid(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
// the forward ends in two fp64 atomics per workgroup on 16 replicas: 2048 workgroups of one row per thread spent most of
// their 14.6 us (3 x 173k rows) queueing on those 32 words; 512 workgroups, four rows per thread there
static inline int ce_grid_fwd(int64_t n) {
  const int g = ce_grid(n);
  return g > 512 ? 512 : g;
}

}  // namespace mtmc

extern "C" {

int32_t mtmc_cross_entropy_forward(const float* logits, const int64_t* labels, const float* weight, int64_t n, int32_t n_classes,
                                   int64_t ignore_index, int32_t mode, float* per_sample, double* sums, float* loss_out,
                                   void* stream) {
  if (!logits || !labels || !sums || n < 0 || n_classes < 1 || n_classes > MTMC_MAX_CLASSES) return MTMC_E_ARG;
  if (n_classes == 2 && ((uintptr_t)logits & 7)) return MTMC_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * mtmc::kStatRep, s) != hipSuccess) return MTMC_E_HIP;
  if (n > 0)
    hipLaunchKernelGGL(mtmc::ce_forward_kernel, dim3(mtmc::ce_grid_fwd(n)), dim3(256), 0, s, logits, labels, weight, n,
                       n_classes, ignore_index, per_sample, sums, (int64_t)0);
  hipLaunchKernelGGL(mtmc::ce_finalize_kernel, dim3(1), dim3(1), 0, s, sums, mode, loss_out, 1.0);
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

// The training loop's `sum(criterion(step, labels) for step in outputs['classified_edges'])` (reference train.py:118-138:
// every classified step against the SAME labels) in one pass over the [n_steps][n][C] logits block the forward emits:
// with equal labels the weight sums of the steps are equal, so sum_s mean_s = n_steps * (sum of all terms / sum of all
// weights).  mode 0 = mean per step (then summed), 1 = sum.
int32_t mtmc_cross_entropy_steps_forward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                         int32_t n_classes, int32_t n_steps, int64_t ignore_index, int32_t mode,
                                         double* sums, float* loss_out, void* stream) {
  if (!logits || !labels || !sums || n < 0 || n_steps < 1 || n_classes < 1 || n_classes > MTMC_MAX_CLASSES) return MTMC_E_ARG;
  if (mode < 0 || mode > 1 || (n_classes == 2 && ((uintptr_t)logits & 7))) return MTMC_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * mtmc::kStatRep, s) != hipSuccess) return MTMC_E_HIP;
  if (n > 0)
    hipLaunchKernelGGL(mtmc::ce_forward_kernel, dim3(mtmc::ce_grid_fwd(n * n_steps)), dim3(256), 0, s, logits, labels, weight,
                       n * n_steps, n_classes, ignore_index, (float*)nullptr, sums, n);
  hipLaunchKernelGGL(mtmc::ce_finalize_kernel, dim3(1), dim3(1), 0, s, sums, mode, loss_out, (double)n_steps);
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

int32_t mtmc_cross_entropy_steps_backward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                          int32_t n_classes, int32_t n_steps, int64_t ignore_index, int32_t mode,
                                          const float* grad, const double* sums, float* d_logits, void* stream) {
  if (!logits || !labels || !grad || !sums || !d_logits || n < 0 || n_steps < 1 || n_classes < 1 ||
      n_classes > MTMC_MAX_CLASSES || mode < 0 || mode > 1)
    return MTMC_E_ARG;
  if (n_classes == 2 && (((uintptr_t)logits & 7) || ((uintptr_t)d_logits & 7))) return MTMC_E_ARG;
  if (n > 0)
    hipLaunchKernelGGL(mtmc::ce_backward_kernel, dim3(mtmc::ce_grid(n * n_steps)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), logits, labels, weight, n * n_steps, n_classes, ignore_index, mode,
                       grad, sums, d_logits, n, (double)n_steps);
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

int32_t mtmc_cross_entropy_backward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                    int32_t n_classes, int64_t ignore_index, int32_t mode, const float* grad,
                                    const double* sums, float* d_logits, void* stream) {
  if (!logits || !labels || !grad || !d_logits || n < 0 || n_classes < 1 || n_classes > MTMC_MAX_CLASSES) return MTMC_E_ARG;
  if (mode < 0 || mode > 2 || (mode == 0 && !sums)) return MTMC_E_ARG;
  if (n_classes == 2 && (((uintptr_t)logits & 7) || ((uintptr_t)d_logits & 7))) return MTMC_E_ARG;
  if (n > 0)
    hipLaunchKernelGGL(mtmc::ce_backward_kernel, dim3(mtmc::ce_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       logits, labels, weight, n, n_classes, ignore_index, mode, grad, sums, d_logits, (int64_t)0, 1.0);
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

int32_t mtmc_edge_confusion(const float* logits, const int64_t* labels, int64_t n, int32_t n_classes, int64_t* counts,
                            void* stream) {
  if (!logits || !labels || !counts || n < 0 || n_classes < 2 || n_classes > MTMC_MAX_CLASSES) return MTMC_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), s) != hipSuccess) return MTMC_E_HIP;
  if (n > 0)
    hipLaunchKernelGGL(mtmc::confusion_kernel, dim3(mtmc::ce_grid(n) > 256 ? 256 : mtmc::ce_grid(n)), dim3(256), 0, s, logits,
                       labels, n, n_classes, reinterpret_cast<unsigned long long*>(counts));
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

}  // extern "C"
