// C-ABI layer (include/mtmc_mpn.h): argument checking, workspace carving and the phase sequence of one
// MOTMPNet.forward (reference models/mpn.py:250-299).  Launch-only: nothing here allocates or synchronises,
// so a caller may capture a forward into a hipGraph.
#include "api_internal.h"

using namespace mtmc_api;

namespace {

// The edge branch (index conversion, degree, edge-encoder moments) and the node-encoder GEMM chain are
// independent until the first round.  With MTMC_F_FORK the edge branch runs on a side stream (fork/join with
// events; legal under stream capture too).  Opt-in: measured on MI355X/ROCm 7.2 the two event hops cost more
// (~+20 us per forward on the 150k-edge S02 graph) than the ~17 us of overlap they buy.
// One side stream + event pair per device, created on first use.
struct Side { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false; };
Side* side_for_current_device() {
  static Side sides[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Side& sd = sides[dev];
  if (!sd.ok) {
    if (hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    sd.ok = true;
  }
  return &sd;
}

// The pipelined layer 0's side stream and events: one set per (device, CALLER STREAM).  Round 4 had one per device: two host
// threads (or two streams) running many-row forwards on one device then recorded and waited on the SAME event objects, and a
// panel GEMM of one could start on the other's `ready[i]`.  A set whose creation fails half-way is destroyed
// again; at most kMaxPipes sets exist (beyond that: the serial layer 0).
constexpr int kMaxPipes = 64;
struct PipeSlot { int dev; hipStream_t caller; SidePipe pipe; };
void destroy_pipe(SidePipe& sp, int n_events) {
  for (int i = 0; i < n_events; ++i) (void)hipEventDestroy(sp.ready[i]);
  if (sp.fork) (void)hipEventDestroy(sp.fork);
  if (sp.stream) (void)hipStreamDestroy(sp.stream);
  sp = SidePipe();
}
SidePipe* pipe_for(hipStream_t caller) {
  static PipeSlot slots[kMaxPipes];
  static int n_slots = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n_slots; ++i)
    if (slots[i].dev == dev && slots[i].caller == caller) return &slots[i].pipe;
  if (n_slots == kMaxPipes) return nullptr;
  SidePipe sp = SidePipe();
  if (hipStreamCreateWithFlags(&sp.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&sp.fork, hipEventDisableTiming) != hipSuccess) { sp.fork = nullptr; destroy_pipe(sp, 0); return nullptr; }
  for (int i = 0; i < kMaxPanels; ++i)
    if (hipEventCreateWithFlags(&sp.ready[i], hipEventDisableTiming) != hipSuccess) { destroy_pipe(sp, i); return nullptr; }
  slots[n_slots] = {dev, caller, sp};
  return &slots[n_slots++].pipe;
}

}  // namespace

extern "C" {

int32_t mtmc_mpn_abi_version(void) { return MTMC_MPN_ABI_VERSION; }
const char* mtmc_mpn_last_error(void) { return g_err; }

size_t mtmc_mpn_train_workspace_bytes(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges) {
  if (check_model(model) != MTMC_OK || n_nodes < 0 || n_edges < 0) return 0;
  Layout lo;
  make_layout(model, n_nodes, n_edges, &lo, true);
  return lo.pub.total_bytes;
}

size_t mtmc_mpn_workspace_bytes(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges) {
  if (check_model(model) != MTMC_OK || n_nodes < 0 || n_edges < 0) return 0;
  Layout lo;
  make_layout(model, n_nodes, n_edges, &lo);
  return lo.pub.total_bytes;
}

size_t mtmc_mpn_weight_cache_bytes(const mtmc_mpn_model* model) {
  if (check_model(model) != MTMC_OK) return 0;
  CacheLayout cl;
  make_cache_layout(model, &cl);
  return cl.total > 0 ? cl.total : 256;
}

int32_t mtmc_mpn_workspace_layout(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges, mtmc_ws_layout* out) {
  if (int rc = check_model(model)) return rc;
  if (!out || n_nodes < 0 || n_edges < 0) return fail(MTMC_E_ARG, "bad layout arguments");
  Layout lo;
  make_layout(model, n_nodes, n_edges, &lo);
  *out = lo.pub;
  return MTMC_OK;
}

// Diagnostics: time every layer-0 GEMM launch (row panels) of the many-row forwards that follow with HIP events on the call's
// stream; mtmc_dbg_panel_times synchronises the LAST forward's events and returns their number (elapsed ms in `ms`).
int32_t mtmc_dbg_panel_timing(int32_t enable) {
  PanelTiming& pt = panel_timing();
  if (enable && !pt.created) {
    for (int i = 0; i < 2 * kMaxPanels; ++i)
      if (hipEventCreate(&pt.ev[i]) != hipSuccess) return fail(MTMC_E_HIP, "hipEventCreate failed");
    pt.created = true;
  }
  pt.on = enable != 0;
  pt.n = 0;
  return MTMC_OK;
}
int32_t mtmc_dbg_panel_times(float* ms, int32_t max) {
  PanelTiming& pt = panel_timing();
  if (!ms || !pt.created) return 0;
  int n = 0;
  for (; n < pt.n && n < max; ++n) {
    if (hipEventSynchronize(pt.ev[2 * n + 1]) != hipSuccess) break;
    if (hipEventElapsedTime(&ms[n], pt.ev[2 * n], pt.ev[2 * n + 1]) != hipSuccess) break;
  }
  return n;
}

int32_t mtmc_mpn_run_phase(const mtmc_mpn_model* model, const mtmc_mpn_call* call, int32_t phase, int32_t arg) {
  Ctx x;
  if (int rc = make_ctx(model, call, &x)) return rc;
  return run_phase(x, phase, arg);
}

int32_t mtmc_mpn_run_phases(const mtmc_mpn_model* model, const mtmc_mpn_call* call, const int32_t* phase_args, int32_t n) {
  Ctx x;
  if (int rc = make_ctx(model, call, &x)) return rc;
  if (n < 0 || (n > 0 && !phase_args)) return fail(MTMC_E_ARG, "mtmc_mpn_run_phases: bad phase list");
  for (int i = 0; i < n; ++i)
    if (int rc = run_phase(x, phase_args[2 * i], phase_args[2 * i + 1])) return rc;
  return MTMC_OK;
}

int32_t mtmc_mpn_plan_call(const mtmc_mpn_model* model, const mtmc_mpn_call* call, mtmc_mpn_plan* out) {
  if (int rc = check_model(model)) return rc;
  if (!out) return fail(MTMC_E_ARG, "mtmc_mpn_plan_call: NULL result");
  if (int rc = check_call_size(call)) return rc;
  const mtmc_mpn_call* c = call;
  if (c->n_nodes < 0 || c->n_edges < 0 || c->n_edges > c->n_edges_total || c->node_lo < 0 || c->node_hi < c->node_lo ||
      c->node_hi > c->n_nodes || c->row_lo < 0 || c->row_hi < c->row_lo || c->row_hi > c->n_nodes)
    return fail(MTMC_E_ARG, "mtmc_mpn_plan_call: sizes / ranges out of order");
  *out = mtmc_mpn_plan();
  const int64_t rows = c->node_hi - c->node_lo;
  const bool pre0 = !c->training && mtmc::presplit_layer0(c->n_nodes, model->enc_node[0].in_dim, model->enc_node[0].out_dim) &&
                    mtmc::presplit_layer0(rows, model->enc_node[0].in_dim, model->enc_node[0].out_dim);
  CacheLayout cl;
  make_cache_layout(model, &cl);
  bool few = c->weight_cache != nullptr && few_shape(model, c->n_nodes) && few_shape(model, rows);   // use_few (training too)
  for (int l = 0; l < model->n_enc_layers; ++l) few = few && cl.has[l];
  for (int l = 0; l < model->n_enc_layers; ++l) {
    const mtmc_layer& L = model->enc_node[l];
    if (few) {
      out->enc_kernel[l] = l == 0 ? MTMC_GEMM_FEW_L0 : MTMC_GEMM_FEW_WAVE;
      out->enc_split_k[l] = 1;
      continue;
    }
    int sk_full = 1, sk_here = 1;
    mtmc::gemm_plan(c->n_nodes, L.in_dim, L.out_dim, &sk_full);
    const int cfg = rows > 0 ? mtmc::gemm_plan(rows, L.in_dim, L.out_dim, &sk_here) : 0;
    const bool slab = sk_here > 1 && (size_t)sk_here * rows <= (size_t)(sk_full > 1 ? sk_full : 0) * c->n_nodes;   // run_phase
    const bool stg = l >= 1 && !c->training && mtmc::staged_layer(c->n_nodes, L.in_dim, L.out_dim) && mtmc::staged_layer(rows, L.in_dim, L.out_dim);
    const bool rws = l >= 1 && !c->training && mtmc::rows_layer(c->n_nodes, L.in_dim, L.out_dim) && mtmc::rows_layer(rows, L.in_dim, L.out_dim);
    out->enc_kernel[l] = (l == 0 && pre0) ? MTMC_GEMM_PRESPLIT_256 : stg ? MTMC_GEMM_STAGED_128 : rws ? MTMC_GEMM_ROWS_16 : cfg == 2 ? MTMC_GEMM_INLOOP_128 : cfg == 1 ? MTMC_GEMM_INLOOP_64
                                                                                                           : MTMC_GEMM_GENERIC;
    out->enc_split_k[l] = ((l == 0 && pre0) || stg || rws || !slab) ? 1 : sk_here;
  }
  out->edges_per_thread = mtmc::plan_edges_per_thread(c->n_edges);
  out->lazy_edges = lazy_edges(c) ? 1 : 0;
  out->avg_degree = avg_degree(c);
  out->pass_a_col_blocks = (model->num_enc_steps > 0 && mtmc::plan_col_blocks(c->n_nodes, c->n_edges, 1e30, c->training != 0) > 0)
                               ? mtmc::plan_col_blocks(c->n_nodes, c->n_edges, out->avg_degree, c->training != 0) : 0;
  const bool drop_n = c->training && model->dropout_upd_node > 0.f;
  // the public enum names what launch_pass_c launches: the sorted kernel is MFMA_SORTED, the any-order kernel MFMA_ANY
  // (on a many-edge list it still has the walk launched behind it for unsorted rows)
  const int pc = mtmc::plan_pass_c(model->agg, (c->flags & MTMC_F_DETERMINISTIC) != 0, drop_n, c->n_edges, c->n_nodes, out->avg_degree);
  out->pass_c = pc == 1 ? (mtmc::pass_c_sorted_taken(c->n_nodes) ? MTMC_PASS_C_MFMA_SORTED : MTMC_PASS_C_MFMA_ANY) : pc;
  out->node_stat_folded = mtmc::fold_node_stat(c->n_edges) ? 1 : 0;
  out->layer0_panels = 1;
  if (pre0 && mtmc::knobs().l0_pipeline > 0) {
    int64_t cuts[kMaxPanels + 1];
    int bm;
    out->layer0_panels = l0_panels(rows, model->enc_node[0].out_dim, cuts, &bm);
  }
  {   // (enc2_can_ride without a Ctx: the same conditions on sizes alone)
    const int last = model->n_enc_layers - 1;
    const mtmc_layer& L = model->enc_node[last];
    int sk;
    const bool big_last = (last == 0 && pre0) ||
                          (last >= 1 && !c->training && ((mtmc::staged_layer(c->n_nodes, L.in_dim, L.out_dim) && mtmc::staged_layer(rows, L.in_dim, L.out_dim)) ||
                                                         (mtmc::rows_layer(c->n_nodes, L.in_dim, L.out_dim) && mtmc::rows_layer(rows, L.in_dim, L.out_dim))));
    out->enc2_passenger = (c->n_edges > 0 && rows > 0 && c->n_edges <= mtmc::kSmallEdges && !big_last &&
                           mtmc::gemm_plan(rows, L.in_dim, L.out_dim, &sk) == 1 && !(c->flags & MTMC_F_FORK)) ? 1 : 0;
    if (few)
      out->enc2_passenger = (c->n_edges > 0 && c->n_edges <= mtmc::kSmallEdges && last >= 1 &&
                             mtmc::few_wave_threads(L.in_dim) == 256 && !(c->flags & MTMC_F_FORK)) ? 1 : 0;
  }
  return MTMC_OK;
}

int32_t mtmc_mpn_forward(const mtmc_mpn_model* model, const mtmc_mpn_call* call) {
  Ctx x;
  if (int rc = make_ctx(model, call, &x)) return rc;
  int rc;
  // many-row graphs: layer 0 in row panels, the operand split one panel ahead on a side stream (MTMC_L0_PIPELINE=0: off)
  if (mtmc::knobs().l0_pipeline > 0 && use_presplit0(x)) {
    int64_t cuts[kMaxPanels + 1];
    int bm;
    if (l0_panels(call->node_hi - call->node_lo, model->enc_node[0].out_dim, cuts, &bm) >= 2) x.pipe = pipe_for(x.stream);
  }
  Side* sd = (call->flags & MTMC_F_FORK) ? side_for_current_device() : nullptr;
  if (sd) {
    if ((rc = run_phase(x, MTMC_PH_BEGIN, 0))) return rc;      // prep also gathers the node encoder's operand scales
    if (hipEventRecord(sd->fork, x.stream) != hipSuccess || hipStreamWaitEvent(sd->stream, sd->fork, 0) != hipSuccess)
      return fail(MTMC_E_HIP, "fork onto the side stream failed");
    Ctx xs = x;
    xs.stream = sd->stream;
    if ((rc = run_phase(xs, MTMC_PH_EDGE_ENC, 0))) return rc;
    if (hipEventRecord(sd->join, sd->stream) != hipSuccess) return fail(MTMC_E_HIP, "hipEventRecord failed");
  } else {
    // (round 5 tried the edge part of prep_kernel as passenger of encoder layer 0, with the statistics head cleared by the operand
    //  jobs' launch instead of a memset: layer 0 took 6 us longer, the jobs alone 8.0 us against prep_kernel's 8.8 -- DESIGN.md A.5)
    if ((rc = run_phase(x, MTMC_PH_BEGIN, 0))) return rc;
    x.enc2_rides = enc2_can_ride(x);              // few-row graphs: enc2 as passenger of the last encoder layer's launch
    if (!x.enc2_rides && (rc = run_phase(x, MTMC_PH_EDGE_ENC, 0))) return rc;
  }
  for (int l = 0; l < model->n_enc_layers; ++l) {
    if ((rc = run_phase(x, MTMC_PH_NODE_ENC, l))) return rc;
    if ((rc = run_phase(x, MTMC_PH_NODE_COMBINE, l))) return rc;
  }
  if (sd && hipStreamWaitEvent(x.stream, sd->join, 0) != hipSuccess) return fail(MTMC_E_HIP, "join failed");
  // single shard: h0 = relu(bn(Y_last)) is produced inside the first round's projection kernel
  const bool fused_h0 = model->num_enc_steps > 0 && call->node_lo == 0 && call->node_hi == call->n_nodes;
  if (!fused_h0 && (rc = run_phase(x, MTMC_PH_NODE_H0, 0))) return rc;
  for (int r = 0; r < model->num_enc_steps; ++r) {
    if ((rc = run_phase(x, MTMC_PH_ROUND_PROJ, r, fused_h0))) return rc;
    if ((rc = run_phase(x, MTMC_PH_ROUND_A, r))) return rc;
    if ((rc = run_phase(x, MTMC_PH_ROUND_B, r))) return rc;
    if ((rc = run_phase(x, MTMC_PH_ROUND_STAT, r))) return rc;
    if ((rc = run_phase(x, MTMC_PH_ROUND_C, r))) return rc;
  }
  return run_phase(x, MTMC_PH_END, 0);
}

static int scatter_common(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                          float* out, float* count, int64_t* arg_out, int mode, void* stream) {
  if (n_src < 0 || n_cols < 1 || dim_size < 0 || !out || (n_src > 0 && (!src || !index)))
    return fail(MTMC_E_ARG, "bad scatter arguments");
  if (mode == 1 && !count) return fail(MTMC_E_ARG, "scatter_mean needs count_scratch[dim_size]");
  if (dim_size == 0) return MTMC_OK;
  mtmc::launch_scatter(src, index, n_src, n_cols, dim_size, out, count, arg_out, mode, static_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "scatter launch failed: %s", hipGetErrorString(e));
}
int32_t mtmc_scatter_add(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                         float* out, void* stream) {
  return scatter_common(src, index, n_src, n_cols, dim_size, out, nullptr, nullptr, 0, stream);
}
int32_t mtmc_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                             int64_t* out, void* stream) {
  if (n_src < 0 || n_cols < 1 || dim_size < 0 || !out || (n_src > 0 && (!src || !index)))
    return fail(MTMC_E_ARG, "bad scatter arguments");
  if (dim_size == 0) return MTMC_OK;
  mtmc::launch_scatter_add_i64(src, index, n_src, n_cols, dim_size, out, static_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "scatter launch failed: %s", hipGetErrorString(e));
}
int32_t mtmc_scatter_mean(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                          float* out, float* count_scratch, void* stream) {
  return scatter_common(src, index, n_src, n_cols, dim_size, out, count_scratch, nullptr, 1, stream);
}
int32_t mtmc_scatter_max(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                         float* out, int64_t* arg_out, void* stream) {
  return scatter_common(src, index, n_src, n_cols, dim_size, out, nullptr, arg_out, 2, stream);
}

int32_t mtmc_mlp_layer_forward(const mtmc_layer* layer, const float* x, int64_t x_row_stride, int64_t rows, float* y,
                               double* stats_scratch, void* stream) {
  if (!layer || !layer->weight || !layer->bias || !x || !y || rows < 1) return fail(MTMC_E_ARG, "bad mlp layer arguments");
  const bool bn = layer->gamma != nullptr;
  if (bn && (!layer->beta || !stats_scratch)) return fail(MTMC_E_ARG, "BatchNorm layer needs beta and stats_scratch[2*out]");
  if (bn && rows < 2) return fail(MTMC_E_ROWS, "Expected more than 1 value per channel when training");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (bn && hipMemsetAsync(stats_scratch, 0, 2 * (size_t)layer->out_dim * sizeof(double), s) != hipSuccess)
    return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  mtmc::GemmParams g;
  g.A = x; g.lda = x_row_stride; g.W = layer->weight; g.bias = layer->bias; g.Y = y; g.ldy = layer->out_dim;
  g.stats_in = nullptr; g.gamma_in = nullptr; g.beta_in = nullptr; g.count = (double)rows;
  g.stats_out = bn ? stats_scratch : nullptr; g.M = rows; g.K = layer->in_dim; g.Nout = layer->out_dim;
  g.slab = nullptr; g.split_k = 1; g.drop_in = {0, 0, 1.f, 0}; g.drop_stream = 0;
  if (mtmc::launch_gemm_bn(g, s) != MTMC_OK) return fail(MTMC_E_ARG, "unsupported layer shape");
  if (bn) {
    mtmc::Drop nodrop = {0, 0, 1.f, 0};
    mtmc::launch_bn_relu_rows(y, layer->out_dim, rows, layer->out_dim, stats_scratch, layer->gamma, layer->beta, (double)rows, y, nodrop, 0, 0, s);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "mlp layer launch failed: %s", hipGetErrorString(e));
}

// Diagnostics / unit tests: Y[M][N] = A[M][K] . W[N][K]^T + bias through the node encoder's GEMM dispatch exactly as
// the forward uses it for layer 0 (operand |.|max gathered by prep_kernel's passenger workgroups, fp16 two-piece
// kernel where it applies).  scratch: u32[48], stats: f64[2*N] or NULL.
int32_t mtmc_linear_raw(const float* A, int64_t lda, const float* W, const float* bias, float* Y, int64_t M, int32_t K,
                        int32_t N, uint32_t* scratch, double* stats, void* stream) {
  if (!A || !W || !bias || !Y || !scratch || M < 1 || K < 32 || K % 32 || N < 1) return fail(MTMC_E_ARG, "bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, 3 * mtmc::kAmaxRep * sizeof(uint32_t), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  mtmc::PrepParams p = {};
  p.n_edges = 0; p.n_jobs = 2;
  p.jobs[0] = {A, M, K, lda, scratch, 0, 0};
  p.jobs[1] = {W, N, K, K, scratch + mtmc::kAmaxRep, 0, 0};
  mtmc::launch_prep(p, s);
  mtmc::GemmParams g;
  g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.Y = Y; g.ldy = N;
  g.stats_in = nullptr; g.gamma_in = nullptr; g.beta_in = nullptr; g.count = (double)M;
  g.stats_out = stats; g.M = M; g.K = K; g.Nout = N;
  g.slab = nullptr; g.split_k = 1; g.drop_in = {0, 0, 1.f, 0}; g.drop_stream = 0;
  g.amax_a = scratch; g.amax_w = scratch + mtmc::kAmaxRep; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
  if (mtmc::launch_gemm_bn(g, s) != MTMC_OK) return fail(MTMC_E_ARG, "unsupported shape");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "launch failed: %s", hipGetErrorString(e));
}

// Diagnostics / unit tests: one encoder layer >= 1 as the forward runs it on many-row graphs:
// Y[M][N] = relu(bn(A))[M][K] . W[N][K]^T + bias, bn = BatchNorm with the given column statistics (f64 sum[K] | sumsq[K] over
// `count` rows) and gamma / beta -- on the role-split kernel (gemm_staged.hip: N % 256 == 0, or N = 128) or, for the narrow last layer
// (K = 128, N = 32), the row-streaming kernel (gemm_rows.hip).  work: >= 4*N*K + 4*N + 256 bytes (the weight planes of the
// role-split kernel); scratch: u32[48]; stats: f64[2*N] or NULL.
int32_t mtmc_linear_staged_raw(const float* A, int64_t lda, const double* stats_in, const float* gamma_in, const float* beta_in,
                               double count, const float* W, const float* bias, float* Y, int64_t M, int32_t K, int32_t N,
                               void* work, uint64_t work_bytes, uint32_t* scratch, double* stats, void* stream) {
  const bool narrow = K == 128 && N == 32;
  if (!A || !stats_in || !gamma_in || !beta_in || !W || !bias || !Y || !work || !scratch || M < 1 || K < 64 || K % 32 || K > 2048 ||
      (!narrow && N != 128 && (N < 256 || N % 256)) || lda < K || (lda & 3) || ((uintptr_t)A & 15))
    return fail(MTMC_E_ARG, "bad arguments");
  const uint64_t iw_off = ((uint64_t)N * K * 4 + 255) / 256 * 256;
  if (work_bytes < iw_off + (uint64_t)N * 4) return fail(MTMC_E_ARG, "work buffer too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, 3 * mtmc::kAmaxRep * sizeof(uint32_t), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  mtmc::PrepParams pp = {};
  pp.n_edges = 0; pp.n_jobs = 1;
  pp.jobs[0] = {A, M, K, lda, scratch, 0, 0};
  mtmc::launch_prep(pp, s);
  int rc;
  if (narrow) {
    mtmc::GemmParams g;
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.Y = Y; g.ldy = N;
    g.stats_in = stats_in; g.gamma_in = gamma_in; g.beta_in = beta_in; g.count = count;
    g.stats_out = stats; g.M = M; g.K = K; g.Nout = N;
    g.slab = nullptr; g.split_k = 1; g.drop_in = {0, 0, 1.f, 0}; g.drop_stream = 0;
    g.amax_a = scratch; g.amax_w = nullptr; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
    rc = mtmc::launch_gemm_rows(g, s);
  } else {
    unsigned char* wk = static_cast<unsigned char*>(work);
    mtmc::launch_split_rows(W, K, N, K, wk, reinterpret_cast<float*>(wk + iw_off), s);
    mtmc::StagedGemmParams g;
    g.A = A; g.lda = lda; g.stats_in = stats_in; g.gamma_in = gamma_in; g.beta_in = beta_in; g.count = count;
    g.amax_a = scratch; g.Wh = reinterpret_cast<const _Float16*>(wk); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
    g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
    g.M = M; g.K = K; g.Nout = N;
    rc = mtmc::launch_gemm_staged(g, s);
  }
  if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "unsupported shape or launch refused");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "launch failed: %s", hipGetErrorString(e));
}

// Diagnostics / unit tests: one encoder layer as the forward runs it on few-row graphs (gemm_few.hip).
int32_t mtmc_linear_few_raw(const float* A, int64_t lda, const double* stats_in, const float* gamma_in, const float* beta_in,
                            double count, const float* W, const float* bias, float* Y, int64_t M, int32_t K, int32_t N,
                            void* work, uint64_t work_bytes, double* stats, void* stream) {
  const bool l0 = stats_in == nullptr;
  if (!A || !W || !bias || !Y || !work || M < 1 || K < 32 || K > 2048 || N < 1 || lda < K || (lda & 3) || ((uintptr_t)A & 15) ||
      (l0 ? !mtmc::few_l0_shape(K, N) : (!mtmc::few_wave_shape(K, N) || !gamma_in || !beta_in)))
    return fail(MTMC_E_ARG, "bad arguments");
  const uint64_t a_bytes = l0 ? (uint64_t)M * K * 4 : 0, w_bytes = (uint64_t)N * K * 4;
  const uint64_t ia_off = a_bytes, wh_off = (ia_off + (uint64_t)M * 4 + 255) / 256 * 256, iw_off = wh_off + w_bytes;
  if (work_bytes < iw_off + (uint64_t)N * 4) return fail(MTMC_E_ARG, "work buffer too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  unsigned char* wk = static_cast<unsigned char*>(work);
  mtmc::launch_split_rows(W, K, N, K, wk + wh_off, reinterpret_cast<float*>(wk + iw_off), s);
  int rc;
  if (l0) {
    mtmc::launch_split_rows(A, lda, M, K, wk, reinterpret_cast<float*>(wk + ia_off), s);
    mtmc::FewL0Params g;
    g.Ah = reinterpret_cast<const _Float16*>(wk); g.inv_a = reinterpret_cast<const float*>(wk + ia_off);
    g.Wh = reinterpret_cast<const _Float16*>(wk + wh_off); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
    g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.M = M; g.K = K; g.Nout = N;
    rc = mtmc::launch_few_l0(g, s);
  } else {
    mtmc::FewWaveParams g;
    g.A = A; g.lda = lda; g.stats_in = stats_in; g.gamma_in = gamma_in; g.beta_in = beta_in; g.count = count;
    g.Wh = reinterpret_cast<const _Float16*>(wk + wh_off); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
    g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.M = M; g.K = K; g.Nout = N;
    rc = mtmc::launch_few_wave(g, s);
  }
  if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "unsupported shape or launch refused");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "launch failed: %s", hipGetErrorString(e));
}

int32_t mtmc_linear_presplit_raw(const float* A, int64_t lda, const float* W, const float* bias, float* Y, int64_t M,
                                 int32_t K, int32_t N, void* work, uint64_t work_bytes, uint32_t* scratch, double* stats,
                                 int32_t reuse_planes, void* stream) {
  if (!A || !W || !bias || !Y || !work || !scratch || M < 1 || K < 64 || K % 64 || K > 2048 || N < 1)
    return fail(MTMC_E_ARG, "bad arguments");
  const uint64_t a_bytes = (uint64_t)M * K * 4, w_bytes = (uint64_t)N * K * 4;
  const uint64_t ia_off = a_bytes, wh_off = (ia_off + (uint64_t)M * 4 + 255) / 256 * 256, iw_off = wh_off + w_bytes;
  if (work_bytes < iw_off + (uint64_t)N * 4) return fail(MTMC_E_ARG, "work buffer too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(scratch, 0, 3 * mtmc::kAmaxRep * sizeof(uint32_t), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  if (stats && hipMemsetAsync(stats, 0, 2 * (size_t)N * sizeof(double), s) != hipSuccess) return fail(MTMC_E_HIP, "hipMemsetAsync failed");
  unsigned char* wk = static_cast<unsigned char*>(work);
  if (!reuse_planes) {
    mtmc::launch_split_rows(A, lda, M, K, wk, reinterpret_cast<float*>(wk + ia_off), s);
    mtmc::launch_split_rows(W, K, N, K, wk + wh_off, reinterpret_cast<float*>(wk + iw_off), s);
  }
  mtmc::SplitGemmParams g;
  g.Ah = reinterpret_cast<const _Float16*>(wk); g.inv_a = reinterpret_cast<const float*>(wk + ia_off);
  g.Wh = reinterpret_cast<const _Float16*>(wk + wh_off); g.inv_w = reinterpret_cast<const float*>(wk + iw_off);
  g.bias = bias; g.Y = Y; g.ldy = N; g.stats_out = stats; g.amax_y = scratch + 2 * mtmc::kAmaxRep;
  g.M = M; g.K = K; g.Nout = N;
  const int rc = mtmc::launch_gemm_presplit(g, s);
  if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "unsupported shape or launch refused");
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MTMC_OK : fail(MTMC_E_HIP, "launch failed: %s", hipGetErrorString(e));
}

}  // extern "C"
