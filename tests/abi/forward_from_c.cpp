// The C ABI used without PyTorch: a plain HIP host program that loads parameters and a graph from a flat file,
// calls mtmc_mpn_forward once and writes the logits and node states back.  tests/test_gpu_abi_from_c.py builds it
// with hipcc, feeds it a model's state and compares the result with the Python module's.
//   file layout (little endian): i64 N, E, L, Cs, F, n_params; then n_params x (i64 count, f32[count]) in
//   mtmc_mpn_model order (enc_node l: W b gamma beta; enc_edge 0,1: W b gamma beta; upd_edge; upd_node; cls W b);
//   then f32 x[N*F], i64 edge_index[2*E] (row then col), f32 edge_attr[E*2].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "mtmc_mpn.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)

static void* upload(const void* src, size_t bytes) {
  void* d = nullptr;
  if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) return nullptr;
  if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s in.bin out.bin [cache]\n", argv[0]); return 1; }
  const bool with_cache = argc == 4;     // a weight-plane cache (ABI v6): few-row graphs then run csrc/gemm_few.hip's kernels
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int64_t hdr[6];
  if (fread(hdr, 8, 6, f) != 6) return 1;
  const int64_t N = hdr[0], E = hdr[1], L = hdr[2], Cs = hdr[3], F = hdr[4], n_params = hdr[5];
  std::vector<float*> dev;
  std::vector<int64_t> counts;
  for (int64_t i = 0; i < n_params; ++i) {
    int64_t cnt;
    if (fread(&cnt, 8, 1, f) != 1) return 1;
    std::vector<float> h(cnt);
    if (fread(h.data(), 4, cnt, f) != (size_t)cnt) return 1;
    dev.push_back(static_cast<float*>(upload(h.data(), cnt * 4)));
    counts.push_back(cnt);
  }
  std::vector<float> x(N * F), attr(E * 2);
  std::vector<int64_t> ei(2 * E);
  if (fread(x.data(), 4, x.size(), f) != x.size() || fread(ei.data(), 8, ei.size(), f) != ei.size() ||
      fread(attr.data(), 4, attr.size(), f) != attr.size()) return 1;
  fclose(f);

  const int dims[5] = {(int)F, 1024, 512, 128, 32};
  mtmc_mpn_model m = {};
  m.struct_bytes = sizeof(m);
  int k = 0;
  m.n_enc_layers = 4;
  auto layer = [&](mtmc_layer& l, int in, int out, bool bn) {
    l.weight = dev[k++]; l.bias = dev[k++];
    l.gamma = bn ? dev[k++] : nullptr; l.beta = bn ? dev[k++] : nullptr;
    l.in_dim = in; l.out_dim = out;
  };
  for (int l = 0; l < 4; ++l) layer(m.enc_node[l], dims[l], dims[l + 1], true);
  layer(m.enc_edge[0], 2, 4, true);
  layer(m.enc_edge[1], 4, 4, true);
  layer(m.upd_edge, 68, 4, true);
  layer(m.upd_node, 36, 32, true);
  layer(m.cls, 4, 2, false);
  if (k != n_params) { fprintf(stderr, "expected %d parameter tensors, file has %lld\n", k, (long long)n_params); return 1; }
  m.agg = MTMC_AGG_SUM; m.num_enc_steps = (int)L; m.num_class_steps = (int)Cs;

  const int64_t n_out = L > 0 ? (Cs < L ? Cs : L) : 1;
  float* d_x = static_cast<float*>(upload(x.data(), x.size() * 4));
  int64_t* d_ei = static_cast<int64_t*>(upload(ei.data(), ei.size() * 8));
  float* d_attr = static_cast<float*>(upload(attr.data(), attr.size() * 4));
  float *d_logits, *d_h;
  void* ws;
  const size_t ws_bytes = mtmc_mpn_workspace_bytes(&m, N, E);
  CHECK(hipMalloc(&d_logits, n_out * E * 2 * 4));
  CHECK(hipMalloc(&d_h, N * 32 * 4));
  CHECK(hipMalloc(&ws, ws_bytes));
  mtmc_mpn_call c = {};
  c.struct_bytes = sizeof(c);
  c.x = d_x; c.x_row_stride = F; c.row = d_ei; c.col = d_ei + E; c.idx_stride = 1; c.edge_attr = d_attr;
  c.n_nodes = N; c.n_edges = E; c.n_edges_total = E; c.node_lo = 0; c.node_hi = N;
  c.logits = d_logits; c.h_out = d_h; c.workspace = ws; c.workspace_bytes = ws_bytes; c.stream = nullptr;
  if (with_cache) {
    c.weight_cache_bytes = mtmc_mpn_weight_cache_bytes(&m);
    CHECK(hipMalloc(&c.weight_cache, c.weight_cache_bytes));
    CHECK(hipMemset(c.weight_cache, 0, c.weight_cache_bytes));          // zero-filled once; the library verifies it per call
    if (mtmc_mpn_forward(&m, &c) != MTMC_OK) { fprintf(stderr, "first forward: %s\n", mtmc_mpn_last_error()); return 3; }
  }
  const int rc = mtmc_mpn_forward(&m, &c);                              // (with a cache: the second call, every chunk verified)
  if (rc != MTMC_OK) { fprintf(stderr, "mtmc_mpn_forward: %d (%s)\n", rc, mtmc_mpn_last_error()); return 3; }
  CHECK(hipDeviceSynchronize());
  std::vector<float> logits(n_out * E * 2), h(N * 32);
  CHECK(hipMemcpy(logits.data(), d_logits, logits.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(h.data(), d_h, h.size() * 4, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 1;
  fwrite(logits.data(), 4, logits.size(), o);
  fwrite(h.data(), 4, h.size(), o);
  fclose(o);
  printf("abi v%d: N=%lld E=%lld L=%lld -> %lld logit sets\n", mtmc_mpn_abi_version(), (long long)N, (long long)E, (long long)L, (long long)n_out);
  return 0;
}
