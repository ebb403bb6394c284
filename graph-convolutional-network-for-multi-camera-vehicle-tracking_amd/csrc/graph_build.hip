// Tracklet-graph construction on gfx950 (SURVEY.md 8(f)-1,2): everything the reference's callers do on the host /
// with O(E) 2048-d gathers between the per-tracklet features and the MPN call (reference inference.py:402-456,
// train.py:316-342):
//     x = F.normalize(feats, p=2, dim=0);  edges = cross-camera cartesian products;  edge labels;
//     edge_attr[e] = [ ||x_r - x_c + 1e-6||_2 ,  1 - cos(x_r, x_c) ]
// The per-edge 2048-d gather (16 KB per edge in the reference) becomes ONE Gram matrix G = X X^T on the fp32
// matrix cores (gemm_bn_kernel) plus an 8-byte-per-edge epilogue:
//     ||a-b+eps||^2 = |a|^2 + |b|^2 - 2 a.b + 2 eps (sum a - sum b) + F eps^2,    cos = a.b / (max(|a|,e) max(|b|,e))
#include "kernels.h"

namespace mtmc {

// column sums of squares over the nodes (fp64): block = 64 columns x 64 rows
__global__ __launch_bounds__(256) void gb_colnorm_kernel(const float* x, int64_t ld, int64_t n, int f, double* colsq) {
  __shared__ double red[4 * 64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * 64;
  double s = 0;
  if (col < f)
    for (int64_t r = r0 + rg; r < r0 + 64 && r < n; r += 4) { const double v = x[r * ld + col]; s += v * v; }
  red[rg * 64 + cl] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < f) unsafeAtomicAdd(colsq + col, red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl]);
}

// x_out = x / max(||col||, 1e-12) (or a copy); per node |x|^2 and sum x of the normalised row (fp64 -> f32)
__global__ __launch_bounds__(256) void gb_normalize_kernel(const float* x, int64_t ld, int64_t n, int f,
                                                           const double* colsq, int l2norm, float* x_out,
                                                           float* row_sq, float* row_sum) {
  __shared__ double red[2 * 4];
  const int64_t r = blockIdx.x;
  double sq = 0, sm = 0;
  for (int c = threadIdx.x; c < f; c += 256) {
    float v = x[r * ld + c];
    if (l2norm) {
      const float nrm = (float)sqrt(colsq[c]);
      v = v / fmaxf(nrm, 1e-12f);
    }
    x_out[r * f + c] = v;
    sq += (double)v * v;
    sm += v;
  }
  sq = wave_sum(sq);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == kWaveSumLane) { red[threadIdx.x >> 6] = sq; red[4 + (threadIdx.x >> 6)] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    row_sq[r] = (float)(red[0] + red[1] + red[2] + red[3]);
    row_sum[r] = (float)(red[4] + red[5] + red[6] + red[7]);
  }
}

struct EdgeBuildParams {
  const int* in_list; const int* in_off;        // nodes of camera c: in_list[in_off[c] .. in_off[c+1])
  const int* out_list; const int64_t* out_off;  // nodes NOT in camera c, ascending
  const int64_t* block_off;                     // first edge of camera c's block; block_off[n_cams] = E
  int n_cams; int64_t n_nodes; int64_t n_edges; int f;
  const float* G; const float* row_sq; const float* row_sum; const int64_t* node_labels;
  int64_t* edge_index;                          // [E][2] (row, col): its .T is the [2,E] view the callers pass on
  float* edge_attr; float* edge_labels;
};

__global__ __launch_bounds__(256) void gb_edges_kernel(EdgeBuildParams p) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < p.n_edges; e += stride) {
    int lo = 0, hi = p.n_cams;                   // camera block containing e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.block_off[mid] <= e) lo = mid; else hi = mid; }
    const int c = lo;
    const int64_t local = e - p.block_off[c];
    const int64_t n_out = p.out_off[c + 1] - p.out_off[c];
    const int row = p.in_list[p.in_off[c] + (int)(local / n_out)];
    const int col = p.out_list[p.out_off[c] + local % n_out];
    reinterpret_cast<longlong2*>(p.edge_index)[e] = make_longlong2(row, col);
    const float g = p.G[(int64_t)row * p.n_nodes + col];
    const float nr = p.row_sq[row], nc = p.row_sq[col];
    const float eps = 1e-6f;                     // F.pairwise_distance eps
    const float d2 = nr + nc - 2.f * g + 2.f * eps * (p.row_sum[row] - p.row_sum[col]) + (float)p.f * eps * eps;
    const float dist = sqrtf(fmaxf(d2, 0.f));
    const float cosv = g / (fmaxf(sqrtf(nr), 1e-8f) * fmaxf(sqrtf(nc), 1e-8f));   // F.cosine_similarity eps
    reinterpret_cast<float2*>(p.edge_attr)[e] = make_float2(dist, 1.f - cosv);
    if (p.edge_labels) p.edge_labels[e] = p.node_labels[row] == p.node_labels[col] ? 1.f : 0.f;
  }
}

}  // namespace mtmc

namespace {
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
struct GLayout { size_t colsq, row_sq, row_sum, G, zeros, slab, total; };
GLayout graph_layout(int64_t n, int f) {
  GLayout l;
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
  l.colsq = take((size_t)f * sizeof(double));
  l.row_sq = take((size_t)n * sizeof(float));
  l.row_sum = take((size_t)n * sizeof(float));
  l.G = take((size_t)n * n * sizeof(float));
  l.zeros = take((size_t)n * sizeof(float));
  int sk = 1;
  mtmc::gemm_plan(n, f, (int)n, &sk);
  l.slab = take(sk > 1 ? (size_t)sk * n * n * sizeof(float) : 0);
  l.total = off;
  return l;
}
}  // namespace

extern "C" {

size_t mtmc_graph_workspace_bytes(int64_t n_nodes, int32_t feat_dim) {
  if (n_nodes < 1 || n_nodes > 46000 || feat_dim < 32) return 0;
  return graph_layout(n_nodes, feat_dim).total;
}

int32_t mtmc_build_graph(const float* feats, int64_t feat_row_stride, int64_t n_nodes, int32_t feat_dim, int32_t l2norm,
                         const int32_t* in_list, const int32_t* in_off, const int32_t* out_list, const int64_t* out_off,
                         const int64_t* block_off, int32_t n_cams, int64_t n_edges, const int64_t* node_labels,
                         float* x_out, int64_t* edge_index_out, float* edge_attr_out, float* edge_labels_out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!feats || !x_out || n_nodes < 1 || n_nodes > 46000 || feat_dim % 32 != 0 || feat_dim < 32 || n_cams < 1) return MTMC_E_ARG;
  if (n_edges > 0 && (!in_list || !in_off || !out_list || !out_off || !block_off || !edge_index_out || !edge_attr_out)) return MTMC_E_ARG;
  if (edge_labels_out && !node_labels) return MTMC_E_ARG;
  if (((uintptr_t)feats & 15) || (feat_row_stride & 3) || ((uintptr_t)x_out & 15) || ((uintptr_t)workspace & 255)) return MTMC_E_ARG;
  const GLayout l = graph_layout(n_nodes, feat_dim);
  if (!workspace || workspace_bytes < l.total) return MTMC_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* colsq = reinterpret_cast<double*>(ws + l.colsq);
  float* row_sq = reinterpret_cast<float*>(ws + l.row_sq);
  float* row_sum = reinterpret_cast<float*>(ws + l.row_sum);
  float* G = reinterpret_cast<float*>(ws + l.G);
  float* zeros = reinterpret_cast<float*>(ws + l.zeros);
  if (hipMemsetAsync(colsq, 0, (size_t)feat_dim * sizeof(double), s) != hipSuccess) return MTMC_E_HIP;
  if (hipMemsetAsync(zeros, 0, (size_t)n_nodes * sizeof(float), s) != hipSuccess) return MTMC_E_HIP;
  if (l2norm)
    hipLaunchKernelGGL(mtmc::gb_colnorm_kernel, dim3((feat_dim + 63) / 64, (unsigned)((n_nodes + 63) / 64)), dim3(256), 0, s,
                       feats, feat_row_stride, n_nodes, feat_dim, colsq);
  hipLaunchKernelGGL(mtmc::gb_normalize_kernel, dim3((unsigned)n_nodes), dim3(256), 0, s, feats, feat_row_stride, n_nodes,
                     feat_dim, colsq, l2norm, x_out, row_sq, row_sum);
  if (n_edges > 0) {
    mtmc::GemmParams g;
    g.A = x_out; g.lda = feat_dim; g.W = x_out; g.bias = zeros; g.Y = G; g.ldy = n_nodes;
    g.stats_in = nullptr; g.gamma_in = nullptr; g.beta_in = nullptr; g.count = 1; g.stats_out = nullptr;
    g.M = n_nodes; g.K = feat_dim; g.Nout = (int)n_nodes; g.drop_in = {0, 0, 1.f, 0}; g.drop_stream = 0;
    g.slab = l.slab != l.total ? reinterpret_cast<float*>(ws + l.slab) : nullptr; g.split_k = 1;
    int sk = 1;
    mtmc::gemm_plan(n_nodes, feat_dim, (int)n_nodes, &sk);
    if (sk <= 1) g.slab = nullptr;
    if (mtmc::launch_gemm_bn(g, s) != MTMC_OK) return MTMC_E_ARG;
    mtmc::EdgeBuildParams p;
    p.in_list = in_list; p.in_off = in_off; p.out_list = out_list; p.out_off = out_off; p.block_off = block_off;
    p.n_cams = n_cams; p.n_nodes = n_nodes; p.n_edges = n_edges; p.f = feat_dim;
    p.G = G; p.row_sq = row_sq; p.row_sum = row_sum; p.node_labels = node_labels;
    p.edge_index = edge_index_out; p.edge_attr = edge_attr_out; p.edge_labels = edge_labels_out;
    const int64_t blocks = (n_edges + 255) / 256;
    hipLaunchKernelGGL(mtmc::gb_edges_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

}  // extern "C"
