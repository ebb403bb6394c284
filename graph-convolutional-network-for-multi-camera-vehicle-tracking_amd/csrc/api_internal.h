// Internals shared by the C-ABI translation units (api.hip: forward + phases, api_train.hip: backward):
// error text, workspace carving, per-phase parameter blocks and the phase launcher.
#pragma once
#include "kernels.h"

#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace mtmc_api {


inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Layout {
  mtmc_ws_layout pub;
  size_t row32, col32, e_buf[2], P, Q, slab, enc_aff, row_start, carry;
  size_t col_sub; int col_blocks;              // column-blocked pass A: int[N][col_blocks + 1] sub-run boundaries (0 blocks: none)
  size_t amax;                                 // u32[1 + 2*MTMC_MAX_ENC_LAYERS][kAmaxRep]: |x|max, -, |Y_l|max (zeroed)
  size_t amax_w;                               // u32[MTMC_MAX_ENC_LAYERS][kAmaxRep]: |W_l|max of the in-loop layers
  size_t Y[MTMC_MAX_ENC_LAYERS];
  size_t stat_enc_layer[MTMC_MAX_ENC_LAYERS];
  // training: every round keeps its own buffers (the workspace is the backward tape) + backward scratch
  bool training;
  std::vector<size_t> z_tr, e_tr, h_tr;       // per round: z1 [E][4], e' [E][4], aggregated h [N][32]
  std::vector<size_t> P_tr, Q_tr;             // per round: the node projections [2][N][4], [N][32]
  size_t xh, inv_a, wh, inv_w; bool presplit0;   // layer 0 on pre-split operands (many-row graphs): fp16 planes + row scales
  bool few;                                    // few-row graphs (gemm_few.hip): xh / inv_a hold the planes of x (W: weight-plane cache)
  size_t wh_l[MTMC_MAX_ENC_LAYERS], inv_w_l[MTMC_MAX_ENC_LAYERS]; bool staged[MTMC_MAX_ENC_LAYERS];   // layers >= 1 on the
                                               // role-split kernel (gemm_staged.hip): the layer's weight planes + row scales
  size_t g_e[2], g_e0, g_h[2], g_h0, g_P, g_Q, g_de2, g_arg;   // gradients wrt e_r, e0, h_r, h0, P, Q; A^T dz2 [E][4]
  size_t bst;                                  // f64[2L+1][kStatRep][kBwdStride] backward statistics blocks
  size_t bwd_zero, bwd_zero_end;               // the range the backward clears with one memset
  size_t gacc;                                 // f32[kGradRep][kGaccN] (train_kernels.h)
  size_t amax_bwd;                             // u32[2][MTMC_MAX_ENC_LAYERS][kAmaxRep] (zeroed with the backward scratch)
  size_t seed_word = 0;
  size_t gA, gB, tA, tB, tW, tX, zeros, bst_n; // node-encoder backward: gradient ping-pong, transposes, 0-bias, column stats
};
constexpr int kBwdStride = 256;                // doubles per replica of the backward statistics scratch

// The weight-plane cache (mtmc_mpn_call::weight_cache; split_body.h): per node-encoder layer the fp16 planes [2][K/32][out][32],
// the inverse row scales [out] and one 64-bit fingerprint per 8-row chunk.  Depends on the model's dimensions only.
struct CacheLayout { size_t planes[MTMC_MAX_ENC_LAYERS], inv[MTMC_MAX_ENC_LAYERS], fp[MTMC_MAX_ENC_LAYERS]; bool has[MTMC_MAX_ENC_LAYERS]; size_t total; };
inline void make_cache_layout(const mtmc_mpn_model* m, CacheLayout* cl) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  for (int l = 0; l < MTMC_MAX_ENC_LAYERS; ++l) {
    cl->has[l] = l < m->n_enc_layers && m->enc_node[l].in_dim % 8 == 0 && m->enc_node[l].in_dim <= 2048;
    cl->planes[l] = cl->inv[l] = cl->fp[l] = 0;
    if (!cl->has[l]) continue;
    const size_t K = m->enc_node[l].in_dim, O = m->enc_node[l].out_dim;
    cl->planes[l] = take(4 * O * K);
    cl->inv[l] = take(4 * O);
    cl->fp[l] = take(8 * ((O + 7) / 8));
  }
  cl->total = off;
}
// the node encoder's shape admits the few-row kernels for `rows` rows (gemm_few.hip; knobs included)
inline bool few_shape(const mtmc_mpn_model* m, int64_t rows) {
  int in_dim[MTMC_MAX_ENC_LAYERS], out_dim[MTMC_MAX_ENC_LAYERS];
  for (int l = 0; l < m->n_enc_layers; ++l) { in_dim[l] = m->enc_node[l].in_dim; out_dim[l] = m->enc_node[l].out_dim; }
  return mtmc::few_rows_path(rows, m->n_enc_layers, in_dim, out_dim);
}

inline int check_model(const mtmc_mpn_model* m) {
  if (!m) return fail(MTMC_E_ARG, "model is NULL");
  if (m->struct_bytes != sizeof(mtmc_mpn_model))
    return fail(MTMC_E_ARG, "mtmc_mpn_model.struct_bytes is %u, this library's struct has %zu bytes (ABI v%d): rebuild the caller against include/mtmc_mpn.h",
                m->struct_bytes, sizeof(mtmc_mpn_model), MTMC_MPN_ABI_VERSION);
  if (m->n_enc_layers < 1 || m->n_enc_layers > MTMC_MAX_ENC_LAYERS) return fail(MTMC_E_ARG, "n_enc_layers out of range");
  for (int l = 0; l < m->n_enc_layers; ++l) {
    const mtmc_layer& L = m->enc_node[l];
    if (!L.weight || !L.bias || !L.gamma || !L.beta) return fail(MTMC_E_ARG, "node encoder layer %d: NULL parameter", l);
    if (L.in_dim % 32 || L.in_dim < 32 || L.out_dim < 1) return fail(MTMC_E_ARG, "node encoder layer %d: in_dim must be a positive multiple of 32", l);
    if (l && L.in_dim != m->enc_node[l - 1].out_dim) return fail(MTMC_E_ARG, "node encoder layer %d: in_dim != previous out_dim", l);
  }
  if (m->enc_node[m->n_enc_layers - 1].out_dim != MTMC_NODE_DIM) return fail(MTMC_E_ARG, "node encoder must end at width %d", MTMC_NODE_DIM);
  if ((m->enc_edge[0].in_dim != 1 && m->enc_edge[0].in_dim != 2) || m->enc_edge[0].out_dim != 4 ||
      m->enc_edge[1].in_dim != 4 || m->enc_edge[1].out_dim != 4)
    return fail(MTMC_E_ARG, "edge encoder must be in(1|2)->4->4");
  const int hn = (m->reattach_nodes ? 2 : 1) * MTMC_NODE_DIM, he = (m->reattach_edges ? 2 : 1) * MTMC_EDGE_DIM;
  if (m->upd_edge.in_dim != 2 * hn + he || m->upd_edge.out_dim != 4) return fail(MTMC_E_ARG, "edge update layer must be [4, %d]", 2 * hn + he);
  if (m->upd_node.in_dim != hn + 4 || m->upd_node.out_dim != MTMC_NODE_DIM) return fail(MTMC_E_ARG, "node update layer must be [32, %d]", hn + 4);
  if (m->cls.in_dim != 4 || m->cls.out_dim < 1 || m->cls.out_dim > MTMC_MAX_CLASSES) return fail(MTMC_E_ARG, "classifier must be [C<=4, 4]");
  const mtmc_layer* small[] = {&m->enc_edge[0], &m->enc_edge[1], &m->upd_edge, &m->upd_node};
  for (const mtmc_layer* L : small)
    if (!L->weight || !L->bias || !L->gamma || !L->beta) return fail(MTMC_E_ARG, "NULL parameter in an edge/update layer");
  if (!m->cls.weight || !m->cls.bias) return fail(MTMC_E_ARG, "NULL classifier parameter");
  if (m->agg < 0 || m->agg > 2) return fail(MTMC_E_ARG, "agg must be MTMC_AGG_SUM/MEAN/MAX");
  if (m->num_enc_steps < 0 || m->num_class_steps < 0) return fail(MTMC_E_ARG, "negative step count");
  return MTMC_OK;
}

inline void make_layout(const mtmc_mpn_model* m, int64_t N, int64_t E, Layout* lo, bool training = false) {
  *lo = Layout();
  lo->training = training;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  const int L = m->num_enc_steps;
  lo->pub.flags_off = take(8 * sizeof(int32_t));
  // the edge branch's two statistics blocks sit right in front of node-encoder layer 0's / layer 1's column statistics: a
  // multi-GPU host all-reduces each pair as ONE message (mtmc_ws_layout, distributed.py)
  lo->pub.stat_attr_off = take((size_t)mtmc::kStatRep * mtmc::kAttrStride * sizeof(double));
  lo->stat_enc_layer[0] = take(2 * (size_t)m->enc_node[0].out_dim * sizeof(double));
  lo->pub.stat_enc2_off = take((size_t)mtmc::kStatRep * mtmc::kEnc2Stride * sizeof(double));
  for (int l = 1; l < m->n_enc_layers; ++l) lo->stat_enc_layer[l] = take(2 * (size_t)m->enc_node[l].out_dim * sizeof(double));
  for (int l = 0; l < m->n_enc_layers; ++l) lo->pub.stat_enc_layer_off[l] = lo->stat_enc_layer[l];
  lo->pub.stat_round_off = take((size_t)(L > 0 ? L : 1) * mtmc::kRoundBlock * sizeof(double));
  lo->pub.deg_off = take((size_t)N * sizeof(int32_t));
  lo->pub.seg_off = take(mtmc::fold_node_stat(E) ? 0 : (size_t)N * 4 * sizeof(double));   // (few-edge lists: none)
  lo->amax = take((size_t)(1 + 2 * MTMC_MAX_ENC_LAYERS) * mtmc::kAmaxRep * sizeof(uint32_t));
  lo->amax_w = take((size_t)MTMC_MAX_ENC_LAYERS * mtmc::kAmaxRep * sizeof(uint32_t));
  lo->pub.zero_bytes = off;
  lo->pub.deg_global_off = take((size_t)N * sizeof(int32_t));
  lo->pub.h0_off = take((size_t)N * 32 * sizeof(float));
  lo->pub.h_acc_off[0] = take((size_t)N * 32 * sizeof(float));
  lo->pub.h_acc_off[1] = take((size_t)N * 32 * sizeof(float));
  lo->enc_aff = take(16 * sizeof(float));
  lo->seed_word = take(sizeof(unsigned long long));     // MTMC_F_SEED_ON_DEVICE: this forward's Dropout seed (stays on the tape)
  lo->row_start = take((size_t)N * sizeof(int32_t));
  lo->carry = take((size_t)((E + 31) / 32) * 2 * 32 * sizeof(float));
  lo->col_blocks = mtmc::plan_col_blocks(N, E, 1e30, training);      // by table size and edge count alone (the call adds the degree)
  lo->col_sub = take(lo->col_blocks > 0 ? (size_t)N * (lo->col_blocks + 1) * sizeof(int32_t) : 0);
  lo->P = take((size_t)N * 8 * sizeof(float));
  lo->pub.P_off = lo->P;
  lo->Q = take((size_t)N * 32 * sizeof(float));
  lo->row32 = take((size_t)E * sizeof(int32_t));
  lo->col32 = take((size_t)E * sizeof(int32_t));
  lo->e_buf[0] = take((size_t)E * 4 * sizeof(float));
  lo->e_buf[1] = take((size_t)E * 4 * sizeof(float));
  for (int l = 0; l < m->n_enc_layers; ++l) lo->Y[l] = take((size_t)N * m->enc_node[l].out_dim * sizeof(float));
  size_t slab = 0;                                 // split-K scratch of the node encoder (few-row graphs only)
  for (int l = 0; l < m->n_enc_layers; ++l) {
    int sk;
    mtmc::gemm_plan(N, m->enc_node[l].in_dim, m->enc_node[l].out_dim, &sk);
    const size_t need = sk > 1 ? (size_t)sk * N * m->enc_node[l].out_dim * sizeof(float) : 0;
    if (need > slab) slab = need;
  }
  lo->slab = take(slab);
  // Layer 0 of a many-row graph runs on operands split ONCE into fp16 pairs (gemm_presplit.hip): planes of x [2][N][K]
  // and of W0 [2][out][K] plus one power-of-two scale per row.  An x-sized region; eval mode only.
  lo->presplit0 = !training && mtmc::presplit_layer0(N, m->enc_node[0].in_dim, m->enc_node[0].out_dim);
  lo->few = few_shape(m, N);                  // (the call also needs a weight-plane cache: use_few; training forwards too)
  if (lo->presplit0 || lo->few) {
    const size_t K0 = m->enc_node[0].in_dim, O0 = m->enc_node[0].out_dim;
    lo->xh = take((size_t)2 * N * K0 * sizeof(uint16_t));
    lo->inv_a = take((size_t)N * sizeof(float));
    if (lo->presplit0) {                       // (without a weight-plane cache the planes of W0 are made per call, here)
      lo->wh = take((size_t)2 * O0 * K0 * sizeof(uint16_t));
      lo->inv_w = take(O0 * sizeof(float));
    }
  }
  for (int l = 1; l < m->n_enc_layers; ++l) {
    lo->staged[l] = !training && mtmc::staged_layer(N, m->enc_node[l].in_dim, m->enc_node[l].out_dim);
    if (lo->staged[l]) {
      lo->wh_l[l] = take((size_t)2 * m->enc_node[l].out_dim * m->enc_node[l].in_dim * sizeof(uint16_t));
      lo->inv_w_l[l] = take((size_t)m->enc_node[l].out_dim * sizeof(float));
    }
  }
  if (training) {
    for (int r = 0; r < L; ++r) {
      lo->z_tr.push_back(take((size_t)E * 4 * sizeof(float)));
      lo->e_tr.push_back(take((size_t)E * 4 * sizeof(float)));
      lo->h_tr.push_back(take((size_t)N * 32 * sizeof(float)));
      lo->P_tr.push_back(r == 0 ? lo->P : take((size_t)N * 8 * sizeof(float)));     // the round's projections stay on the
      lo->Q_tr.push_back(r == 0 ? lo->Q : take((size_t)N * 32 * sizeof(float)));    // tape: the backward reads them as is
    }
    // everything the backward accumulates into, contiguous: ONE memset clears it (api_train.hip)
    size_t maxd = 0, bn_stats = 0;
    for (int l = 0; l < m->n_enc_layers; ++l) {
      maxd = std::max(maxd, (size_t)m->enc_node[l].in_dim);
      maxd = std::max(maxd, (size_t)m->enc_node[l].out_dim);
      bn_stats += 2 * (size_t)m->enc_node[l].out_dim;
    }
    lo->bwd_zero = off;
    lo->bst = take((size_t)(2 * L + 1) * mtmc::kStatRep * kBwdStride * sizeof(double));   // per round: node, edge; + encoder
    lo->bst_n = take(bn_stats * sizeof(double));
    lo->gacc = take((size_t)16 * 256 * sizeof(float));                   // [kGradRep][kGaccN] small-gradient replicas
    lo->amax_bwd = take((size_t)2 * MTMC_MAX_ENC_LAYERS * mtmc::kAmaxRep * sizeof(uint32_t));   // |dY_l|max, |a_{l-1}|max
    lo->g_P = take((size_t)(L > 0 ? L : 1) * 16 * N * 8 * sizeof(float)); // per round: [kGradRep = 16][N][8]
    lo->g_Q = take((size_t)(L > 0 ? L : 1) * N * 32 * sizeof(float));     // per round
    lo->zeros = take(maxd * sizeof(float));
    lo->g_h0 = take((size_t)N * 32 * sizeof(float));
    lo->g_e0 = take((size_t)E * 4 * sizeof(float));
    lo->g_e[0] = take((size_t)E * 4 * sizeof(float));
    lo->g_h[0] = take((size_t)N * 32 * sizeof(float));
    lo->bwd_zero_end = off;
    lo->g_e[1] = take((size_t)E * 4 * sizeof(float));
    lo->g_h[1] = take((size_t)N * 32 * sizeof(float));
    lo->g_de2 = take((size_t)E * 4 * sizeof(float));
    lo->g_arg = take((size_t)N * 32 * sizeof(int32_t));
    const size_t npad = (size_t)((N + 31) / 32 * 32);
    lo->gA = take((size_t)N * maxd * sizeof(float));
    lo->gB = take((size_t)N * maxd * sizeof(float));
    lo->tA = take(npad * maxd * sizeof(float));
    lo->tB = take(npad * maxd * sizeof(float));
    size_t wsum = 0;                                   // every W_l^T and x^T at once: one transpose launch (api_train.hip)
    for (int l = 0; l < m->n_enc_layers; ++l) wsum += (size_t)m->enc_node[l].in_dim * m->enc_node[l].out_dim;
    lo->tW = take(wsum * sizeof(float));
    lo->tX = take(npad * (size_t)m->enc_node[0].in_dim * sizeof(float));
  }
  lo->pub.total_bytes = off;
}

// Side stream + events of the pipelined layer 0 (api.hip: one set per device, created on first use)
constexpr int kMaxPanels = 16;
struct SidePipe { hipStream_t stream = nullptr; hipEvent_t fork = nullptr; hipEvent_t ready[kMaxPanels] = {}; };

// Diagnostics (bench.py): HIP events around every layer-0 GEMM launch of a many-row forward -- in the pipelined forward the
// layer is several panel launches beside the operand split on a side stream, and the figure a bench line reports must be
// that of the forward it sits beside.  Off by default; mtmc_dbg_panel_timing / mtmc_dbg_panel_times (api.hip).
struct PanelTiming { bool on = false, created = false; hipEvent_t ev[2 * kMaxPanels]; int n = 0; };
inline PanelTiming& panel_timing() { static PanelTiming t; return t; }

struct Ctx {
  const mtmc_mpn_model* m;
  const mtmc_mpn_call* c;
  Layout lo;
  char* ws;
  hipStream_t stream;
  char* wc = nullptr;                // the weight-plane cache of an eval-mode call, or nullptr (training / none given)
  CacheLayout cl;
  template <typename T> T* wc_at(size_t off) const { return reinterpret_cast<T*>(wc + off); }
  const SidePipe* pipe = nullptr;    // set by mtmc_mpn_forward: layer 0 runs in row panels (split of panel i+1 beside GEMM i)
  bool enc2_rides = false;           // set by mtmc_mpn_forward (few-row graphs): MTMC_PH_EDGE_ENC's work rides as passenger
                                     // workgroups in the last node-encoder layer's GEMM launch instead of a launch of its own
  template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
};

// the size guard of the call struct (ABI v5): a caller built against an older, shorter struct is refused, not read past
inline int check_call_size(const mtmc_mpn_call* c) {
  if (!c) return fail(MTMC_E_ARG, "call is NULL");
  if (c->struct_bytes != sizeof(mtmc_mpn_call))
    return fail(MTMC_E_ARG, "mtmc_mpn_call.struct_bytes is %u, this library's struct has %zu bytes (ABI v%d): rebuild the caller against include/mtmc_mpn.h",
                c->struct_bytes, sizeof(mtmc_mpn_call), MTMC_MPN_ABI_VERSION);
  return MTMC_OK;
}

inline int make_ctx(const mtmc_mpn_model* m, const mtmc_mpn_call* c, Ctx* ctx) {
  if (int rc = check_model(m)) return rc;
  if (int rc = check_call_size(c)) return rc;
  if (c->training && (c->node_lo != 0 || c->node_hi != c->n_nodes || c->n_edges_total != c->n_edges))
    return fail(MTMC_E_ARG, "training mode is single-shard only");
  if (c->training && (c->flags & MTMC_F_SEED_ON_DEVICE) && (c->seed == 0 || (c->seed & 7)))
    return fail(MTMC_E_ARG, "MTMC_F_SEED_ON_DEVICE: seed must hold the 8-byte aligned device address of a uint64 counter");
  if (c->n_nodes < 2) return fail(MTMC_E_ROWS, "BatchNorm over %lld node rows: Expected more than 1 value per channel", (long long)c->n_nodes);
  if (c->n_edges_total < 2) return fail(MTMC_E_ROWS, "BatchNorm over %lld edge rows: Expected more than 1 value per channel", (long long)c->n_edges_total);
  if (c->n_edges < 0 || c->n_edges > c->n_edges_total || c->n_edges >= (1ll << 31) || c->n_nodes >= (1ll << 31))
    return fail(MTMC_E_ARG, "edge/node counts out of range");
  if (c->node_lo < 0 || c->node_hi < c->node_lo || c->node_hi > c->n_nodes) return fail(MTMC_E_ARG, "bad node range");
  if (c->row_lo < 0 || c->row_hi < c->row_lo || c->row_hi > c->n_nodes) return fail(MTMC_E_ARG, "bad row range");
  if (c->training && (c->row_lo != 0 || (c->row_hi != 0 && c->row_hi != c->n_nodes)))
    return fail(MTMC_E_ARG, "training calls project every node row (row_lo = row_hi = 0)");
  if (!c->x || !c->edge_attr || !c->logits || !c->h_out || !c->workspace) return fail(MTMC_E_ARG, "NULL tensor pointer");
  if (c->n_edges > 0 && (!c->row || !c->col)) return fail(MTMC_E_ARG, "NULL edge_index pointer");
  if (c->idx_stride < 1) return fail(MTMC_E_ARG, "idx_stride must be >= 1");
  if (((uintptr_t)c->x & 15) || (c->x_row_stride & 3) || c->x_row_stride < m->enc_node[0].in_dim)
    return fail(MTMC_E_ARG, "x must be 16-byte aligned with a row stride that is a multiple of 4 and >= in_dim");
  if (((uintptr_t)c->edge_attr & 7) || ((uintptr_t)c->logits & 7) || ((uintptr_t)c->h_out & 15))
    return fail(MTMC_E_ARG, "edge_attr/logits must be 8-byte and h_out 16-byte aligned");
  if ((uintptr_t)c->workspace & 255) return fail(MTMC_E_WORKSPACE, "workspace must be 256-byte aligned");
  ctx->m = m; ctx->c = c;
  make_layout(m, c->n_nodes, c->n_edges, &ctx->lo, c->training != 0);
  if (c->workspace_bytes < ctx->lo.pub.total_bytes)
    return fail(MTMC_E_WORKSPACE, "workspace has %zu bytes, %zu needed", c->workspace_bytes, ctx->lo.pub.total_bytes);
  ctx->ws = static_cast<char*>(c->workspace);
  ctx->stream = static_cast<hipStream_t>(c->stream);
  make_cache_layout(m, &ctx->cl);
  if (c->weight_cache) {
    if ((uintptr_t)c->weight_cache & 255) return fail(MTMC_E_WORKSPACE, "weight_cache must be 256-byte aligned");
    if (c->weight_cache_bytes < ctx->cl.total)
      return fail(MTMC_E_WORKSPACE, "weight_cache has %zu bytes, %zu needed (mtmc_mpn_weight_cache_bytes)", c->weight_cache_bytes, ctx->cl.total);
    ctx->wc = static_cast<char*>(c->weight_cache);
  }
  return MTMC_OK;
}

inline mtmc::Drop make_drop(const Ctx& x, float p) {
  mtmc::Drop d;
  d.on = (x.c->training && p > 0.f) ? 1 : 0;
  d.seed = x.c->seed;
  if (d.on && (x.c->flags & MTMC_F_SEED_ON_DEVICE)) {   // the seed word of this forward, on the tape (seed_tick_kernel)
    d.on = 2;
    d.seed = (unsigned long long)(uintptr_t)(x.ws + x.lo.seed_word);
  }
  const double t = (double)p * 65536.0;                 // 16-bit fields: four elements per 64-bit hash (common.h)
  d.thresh = t >= 65535.0 ? 65535u : (unsigned)t;
  d.inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
  return d;
}

inline mtmc::EdgeEncParams enc_params(const Ctx& x) {
  mtmc::EdgeEncParams e;
  const mtmc_layer& a = x.m->enc_edge[0];
  const mtmc_layer& b = x.m->enc_edge[1];
  e.w1 = a.weight; e.b1 = a.bias; e.g1 = a.gamma; e.bt1 = a.beta;
  e.w2 = b.weight; e.b2 = b.bias; e.g2 = b.gamma; e.bt2 = b.beta;
  e.stat_attr = x.at<double>(x.lo.pub.stat_attr_off);
  e.stat_enc2 = x.at<double>(x.lo.pub.stat_enc2_off);
  e.aff = x.at<float>(x.lo.enc_aff);
  e.fe = a.in_dim;
  e.drop = make_drop(x, x.m->dropout_enc);
  return e;
}

// Where round r aggregates: the last round of a sum/max model writes latent_node_feats straight into h_out.
inline float* agg_target(const Ctx& x, int r) {
  if (x.lo.training) return x.at<float>(x.lo.h_tr[r]);
  if (r == x.m->num_enc_steps - 1 && x.m->agg != MTMC_AGG_MEAN) return x.c->h_out;
  return x.at<float>(x.lo.pub.h_acc_off[r & 1]);
}
// the (unscaled) node state round r reads: h0 for the first round, else what round r-1 aggregated
// rows whose projections / node-update statistics this call computes (mtmc_mpn_call::row_lo / row_hi; 0,0 = all)
inline int64_t proj_lo(const mtmc_mpn_call* c) { return c->row_hi > 0 ? c->row_lo : 0; }
inline int64_t proj_hi(const mtmc_mpn_call* c) { return c->row_hi > 0 ? c->row_hi : c->n_nodes; }
inline float* round_h_src(const Ctx& x, int r) {
  if (r == 0) return x.at<float>(x.lo.pub.h0_off);
  return x.lo.training ? x.at<float>(x.lo.h_tr[r - 1]) : x.at<float>(x.lo.pub.h_acc_off[(r - 1) & 1]);
}
inline float* round_P(const Ctx& x, int r) { return x.at<float>(x.lo.training ? x.lo.P_tr[r] : x.lo.P); }
inline float* round_Q(const Ctx& x, int r) { return x.at<float>(x.lo.training ? x.lo.Q_tr[r] : x.lo.Q); }
inline float* round_z(const Ctx& x, int r) { return x.at<float>(x.lo.training ? x.lo.z_tr[r] : x.lo.e_buf[r & 1]); }
inline float* round_e(const Ctx& x, int r) { return x.at<float>(x.lo.training ? x.lo.e_tr[r] : x.lo.e_buf[r & 1]); }

// edges per source row of this call's edges -- what the pass-C dispatch needs.  Row-complete shard: local edges over the
// rows the call owns; anything else (one GPU, or an edge-range shard that shares rows with its neighbours): the whole
// graph's E / N.  (Round 2 divided the LOCAL edge count by the GLOBAL node count: 8 ranks of config 5 saw 12 instead of
// 100 and left the matrix-core kernel.)
inline double avg_degree(const mtmc_mpn_call* c) {
  if (c->row_hi > 0) return c->row_hi > c->row_lo ? (double)c->n_edges / (double)(c->row_hi - c->row_lo) : 0.0;
  return c->n_nodes > 0 ? (double)c->n_edges_total / (double)c->n_nodes : 0.0;
}
// eval mode, many local edges: e' is never stored (passes B/C and the next round's pass A recompute it from z1)
inline bool lazy_edges(const mtmc_mpn_call* c) { return !c->training && c->n_edges > mtmc::kSmallEdges; }

// column blocks of this call's pass A: the layout has the index (table size, edge count) AND the call's own degree pays
inline int call_col_blocks(const Ctx& x) {
  if (x.lo.col_blocks <= 0) return 0;
  return mtmc::plan_col_blocks(x.c->n_nodes, x.c->n_edges, avg_degree(x.c), x.c->training != 0) == x.lo.col_blocks ? x.lo.col_blocks : 0;
}

inline mtmc::RoundParams round_params(const Ctx& x, int r) {
  const mtmc_mpn_model* m = x.m;
  const int hn = (m->reattach_nodes ? 2 : 1) * MTMC_NODE_DIM;
  mtmc::RoundParams p;
  p.row32 = x.at<int>(x.lo.row32); p.col32 = x.at<int>(x.lo.col32);
  p.attr = x.c->edge_attr;
  p.e_buf = round_z(x, r); p.e_out = round_e(x, r); p.e_prev = r > 0 ? round_e(x, r - 1) : nullptr;
  p.drop_e = make_drop(x, m->dropout_upd_edge); p.drop_n = make_drop(x, m->dropout_upd_node);
  p.drop_stream = mtmc::kDropRound + 2 * r;
  p.P = round_P(x, r); p.Q = round_Q(x, r);
  p.seg = x.at<double>(x.lo.pub.seg_off); p.fold_z2 = mtmc::fold_node_stat(x.c->n_edges) ? 1 : 0;
  p.ue_w = m->upd_edge.weight; p.ue_b = m->upd_edge.bias; p.ue_g = m->upd_edge.gamma; p.ue_bt = m->upd_edge.beta;
  p.ue_ld = m->upd_edge.in_dim; p.ue_eoff = 2 * hn;
  p.un_w = m->upd_node.weight; p.un_b = m->upd_node.bias; p.un_g = m->upd_node.gamma; p.un_bt = m->upd_node.beta;
  p.un_ld = m->upd_node.in_dim; p.un_eoff = hn;
  p.cls_w = m->cls.weight; p.cls_b = m->cls.bias; p.n_classes = m->cls.out_dim;
  p.stats = x.at<double>(x.lo.pub.stat_round_off) + (size_t)r * mtmc::kRoundBlock;
  // (few-edge graphs are latency-bound: there the extra statistics gather in two prologues costs more than the bytes save)
  p.lazy_e = lazy_edges(x.c) ? 1 : 0;
  p.prev_stats = r > 0 ? p.stats - mtmc::kRoundBlock : nullptr;
  p.h_acc = agg_target(x, r);
  const int step = r + 1;
  int first_cls = m->num_enc_steps - m->num_class_steps + 1;   // mpn.py:277; Cs > L classifies every round
  if (first_cls < 1) first_cls = 1;
  p.logits = step >= first_cls ? x.c->logits + (size_t)(step - first_cls) * x.c->n_edges * m->cls.out_dim : nullptr;
  p.n_edges = x.c->n_edges; p.e_total = (double)x.c->n_edges_total;
  p.first_round = r == 0; p.reattach_edges = m->reattach_edges; p.agg = m->agg;
  p.det = (x.c->flags & MTMC_F_DETERMINISTIC) ? 1 : 0; p.flags = x.at<int>(x.lo.pub.flags_off);
  p.deg = x.at<int>(x.lo.pub.deg_off); p.row_start = x.at<int>(x.lo.row_start); p.carry = x.at<float>(x.lo.carry);
  p.n_nodes = x.c->n_nodes;
  p.avg_degree = avg_degree(x.c);
  p.col_blocks = call_col_blocks(x);
  p.col_sub = x.at<int>(x.lo.col_sub); p.cb_row_lo = proj_lo(x.c); p.cb_row_hi = proj_hi(x.c);
  p.enc = enc_params(x);
  return p;
}

// layer 0 on pre-split operands: the layout has the planes (whole graph many-row) AND this call's rows are many too
inline bool use_presplit0(const Ctx& x) {
  return x.lo.presplit0 && mtmc::presplit_layer0(x.c->node_hi - x.c->node_lo, x.m->enc_node[0].in_dim, x.m->enc_node[0].out_dim);
}

// layer l >= 1 on the role-split kernel: the layout has its weight planes AND this call's rows are many too
inline bool use_staged(const Ctx& x, int l) {
  return l >= 1 && x.lo.staged[l] && mtmc::staged_layer(x.c->node_hi - x.c->node_lo, x.m->enc_node[l].in_dim, x.m->enc_node[l].out_dim);
}

// layer l >= 1 on the row-streaming kernel (narrow last layers of many-row graphs, eval mode)
inline bool use_rows(const Ctx& x, int l) {
  return l >= 1 && !x.lo.training && mtmc::rows_layer(x.c->n_nodes, x.m->enc_node[l].in_dim, x.m->enc_node[l].out_dim) &&
         mtmc::rows_layer(x.c->node_hi - x.c->node_lo, x.m->enc_node[l].in_dim, x.m->enc_node[l].out_dim);
}

// every layer on the few-row kernels (gemm_few.hip): eval mode, a weight-plane cache, few rows here AND in the whole graph
inline bool use_few(const Ctx& x) {
  if (!x.lo.few || !x.wc) return false;
  for (int l = 0; l < x.m->n_enc_layers; ++l) if (!x.cl.has[l]) return false;
  return few_shape(x.m, x.c->node_hi - x.c->node_lo);
}
// where layer l's weight planes / inverse row scales are: the cache when the call has one, else the workspace (made per call)
inline _Float16* w_planes(const Ctx& x, int l) {
  if (x.wc && x.cl.has[l]) return x.wc_at<_Float16>(x.cl.planes[l]);
  return x.at<_Float16>(l == 0 ? x.lo.wh : x.lo.wh_l[l]);
}
inline float* w_inv(const Ctx& x, int l) {
  if (x.wc && x.cl.has[l]) return x.wc_at<float>(x.cl.inv[l]);
  return x.at<float>(l == 0 ? x.lo.inv_w : x.lo.inv_w_l[l]);
}

// Row panels of the pipelined layer 0: cuts[0..np] (multiples of the tile height), np <= kMaxPanels; *bm = the tile height
// of every panel (what the whole matrix would pick).  One round = 256 workgroups = 256 / tiles_n row tiles.
inline int l0_panels(int64_t rows, int Nout, int64_t* cuts, int* bm) {
  const int tiles_n = (Nout + 255) / 256;
  *bm = mtmc::presplit_tile_rows(rows, tiles_n);
  const int64_t per_round = (int64_t)(256 / tiles_n > 0 ? 256 / tiles_n : 1) * *bm;
  const int mode = mtmc::knobs().l0_pipeline;
  int np = 0;
  int64_t at = 0, k = 1;
  cuts[0] = 0;
  while (at < rows) {
    int64_t take = per_round * (mode == 3 ? 2 : mode == 2 ? 1 : k);
    if (np == kMaxPanels - 1 || at + take > rows) take = rows - at;
    at += take;
    cuts[++np] = at;
    if (k < 8) k *= 2;
  }
  return np;
}

// Few-row graphs: the last node-encoder layer is a handful of workgroups (S02: 8) on the in-loop kernel, and the edge branch's
// second kernel (enc2: moments of the edge encoder's hidden layer, needed from the first round on) is independent of the whole
// encoder chain -- it rides in that launch as passenger workgroups (GemmParams::pass_*; -1 launch per forward).
inline bool enc2_can_ride(const Ctx& x) {
  const int last = x.m->n_enc_layers - 1;
  const int64_t rows = x.c->node_hi - x.c->node_lo;
  if (x.c->n_edges <= 0 || rows <= 0 || x.c->n_edges > mtmc::kSmallEdges) return false;
  if (use_few(x)) return last >= 1 && mtmc::few_wave_threads(x.m->enc_node[last].in_dim) == 256;   // (enc2_body: 256 threads)
  if ((last == 0 && use_presplit0(x)) || use_staged(x, last) || use_rows(x, last)) return false;
  int sk;
  return mtmc::gemm_plan(rows, x.m->enc_node[last].in_dim, x.m->enc_node[last].out_dim, &sk) == 1;
}

enum { kPhMemset = -1, kPhPrep = -2 };   // the two halves of MTMC_PH_BEGIN, for the forked forward

// the edge part of prep_kernel for this call
inline void fill_prep_edge(const Ctx& x, mtmc::PrepEdge* p) {
  const mtmc_mpn_call* c = x.c;
  p->row = c->row; p->col = c->col; p->idx_stride = c->idx_stride; p->attr = c->edge_attr; p->fe = x.m->enc_edge[0].in_dim;
  p->n_edges = c->n_edges; p->n_nodes = c->n_nodes;
  p->row32 = x.at<int>(x.lo.row32); p->col32 = x.at<int>(x.lo.col32); p->deg = x.at<int>(x.lo.pub.deg_off);
  p->flags = x.at<int>(x.lo.pub.flags_off); p->stat_attr = x.at<double>(x.lo.pub.stat_attr_off);
  p->row_start = x.at<int>(x.lo.row_start);
}

inline const int* scale_deg(const Ctx& x) {   // the degree mean aggregation divides by
  return x.at<int>((x.c->flags & MTMC_F_GLOBAL_DEG) ? x.lo.pub.deg_global_off : x.lo.pub.deg_off);
}

inline int run_phase(const Ctx& x, int phase, int arg, bool fused_h0 = false) {
  const mtmc_mpn_model* m = x.m;
  const mtmc_mpn_call* c = x.c;
  const int L = m->num_enc_steps;
  hipStream_t s = x.stream;
  switch (phase) {
    case MTMC_PH_BEGIN:
    case kPhMemset:
    case kPhPrep: {
      // What depends on the WEIGHTS alone -- the fp16 planes + row scales of the layers that run on pre-split weights -- lives in
      // the call's weight-plane cache when it has one and is VERIFIED here on every call, chunk by chunk, by passenger
      // workgroups of prep_kernel (split_body.h); without a cache it is made per call in the workspace.
      if (phase != kPhPrep && hipMemsetAsync(x.ws, 0, x.lo.pub.zero_bytes, s) != hipSuccess)
        return fail(MTMC_E_HIP, "hipMemsetAsync failed");
      if (phase != kPhPrep && c->training && (c->flags & MTMC_F_SEED_ON_DEVICE))      // seed word <- counter++ (one thread)
        mtmc::launch_seed_tick(reinterpret_cast<unsigned long long*>((uintptr_t)c->seed), x.at<unsigned long long>(x.lo.seed_word), s);
      if (phase != kPhMemset) {
        mtmc::PrepParams p;
        fill_prep_edge(x, &p);
        // passenger jobs: operand |.|max values / operand splits of the node encoder (this rank's rows of x; the weights)
        unsigned* amax = x.at<unsigned>(x.lo.amax);
        p.n_jobs = 0;
        const bool pre0 = use_presplit0(x), few = use_few(x);
        const int64_t rows = c->node_hi - c->node_lo;
        // (training forwards: the jobs also publish the tensors' |.|max -- the backward's GEMMs scale by |x|max and |W_l|max)
        unsigned* amax_w0 = x.at<unsigned>(x.lo.amax_w);
        auto cache_job = [&](int l) {     // layer l's planes in the cache: verify every 8-row chunk, split the ones that changed
          p.jobs[p.n_jobs++] = {m->enc_node[l].weight, m->enc_node[l].out_dim, m->enc_node[l].in_dim, m->enc_node[l].in_dim,
                                c->training ? amax_w0 + l * mtmc::kAmaxRep : nullptr, 0, 0,
                                mtmc::kJobSplit, x.wc_at<_Float16>(x.cl.planes[l]), x.wc_at<float>(x.cl.inv[l]),
                                x.wc_at<unsigned long long>(x.cl.fp[l])};
        };
        if (rows > 0) {
          if (few)         // the planes of x, 8 rows per passenger workgroup (no |x|max: one scale per row)
            p.jobs[p.n_jobs++] = {c->x, rows, m->enc_node[0].in_dim, c->x_row_stride, c->training ? amax : nullptr, 0, 0, mtmc::kJobSplit,
                                  x.at<_Float16>(x.lo.xh), x.at<float>(x.lo.inv_a), nullptr};
          else if (!pre0)
            p.jobs[p.n_jobs++] = {c->x, rows, m->enc_node[0].in_dim, c->x_row_stride, amax, 0, 0, mtmc::kJobAmax, nullptr, nullptr, nullptr};
          for (int l = 0; l < m->n_enc_layers; ++l) {
            const bool planes = few || (l == 0 ? pre0 : use_staged(x, l));
            if (planes) {
              if (x.wc && x.cl.has[l]) cache_job(l);            // (no cache: launch_split_rows below)
            } else if (!use_rows(x, l)) {                       // in-loop kernel: |W_l|max (row-streaming: in the kernel)
              p.jobs[p.n_jobs++] = {m->enc_node[l].weight, m->enc_node[l].out_dim, m->enc_node[l].in_dim, m->enc_node[l].in_dim,
                                    x.at<unsigned>(x.lo.amax_w) + l * mtmc::kAmaxRep, 0, 0, mtmc::kJobAmax, nullptr, nullptr, nullptr};
            }
          }
        }
        mtmc::launch_prep(p, s);
        if (L > 0 && c->n_edges > 0 && call_col_blocks(x) > 0) {   // sub-run boundaries of the column-blocked pass A, once per forward
          const mtmc::RoundParams rp = round_params(x, 0);
          mtmc::launch_colblock_index(rp, x.at<int>(x.lo.col_sub), rp.col_blocks, rp.cb_row_lo, rp.cb_row_hi, s);
        }
        if (pre0) {     // instead of the |.|max of x and W0: their fp16 planes and row scales (one pass over each)
          if (!x.pipe)  // (pipelined layer 0: the x planes are made panel by panel in MTMC_PH_NODE_ENC 0)
            mtmc::launch_split_rows(c->x, c->x_row_stride, rows, m->enc_node[0].in_dim, x.at<void>(x.lo.xh),
                                    x.at<float>(x.lo.inv_a), s);
          if (!(x.wc && x.cl.has[0]))
            mtmc::launch_split_rows(m->enc_node[0].weight, m->enc_node[0].in_dim, m->enc_node[0].out_dim,
                                    m->enc_node[0].in_dim, x.at<void>(x.lo.wh), x.at<float>(x.lo.inv_w), s);
        }
        if (rows > 0)
          for (int l = 1; l < m->n_enc_layers; ++l)
            if (use_staged(x, l) && !(x.wc && x.cl.has[l]))
              mtmc::launch_split_rows(m->enc_node[l].weight, m->enc_node[l].in_dim, m->enc_node[l].out_dim,
                                      m->enc_node[l].in_dim, x.at<void>(x.lo.wh_l[l]), x.at<float>(x.lo.inv_w_l[l]), s);
      }
      break;
    }
    case MTMC_PH_EDGE_ENC:
      if (c->n_edges > 0)
        mtmc::launch_enc2(enc_params(x), c->edge_attr, c->n_edges, (double)c->n_edges_total, x.at<double>(x.lo.pub.stat_enc2_off), s);
      break;
    case MTMC_PH_NODE_ENC:
    case MTMC_PH_NODE_COMBINE: {
      if (arg < 0 || arg >= m->n_enc_layers) return fail(MTMC_E_ARG, "encoder layer %d out of range", arg);
      const int64_t rows = c->node_hi - c->node_lo;
      if (rows == 0) break;
      const mtmc_layer& Lr = m->enc_node[arg];
      if (use_few(x)) {                                          // few-row graphs: one launch per layer, never split along K
        if (phase == MTMC_PH_NODE_COMBINE) break;
        int rc;
        if (arg == 0) {
          mtmc::FewL0Params q;
          q.Ah = x.at<_Float16>(x.lo.xh); q.inv_a = x.at<float>(x.lo.inv_a);
          q.Wh = w_planes(x, 0); q.inv_w = w_inv(x, 0);
          q.bias = Lr.bias; q.Y = x.at<float>(x.lo.Y[0]); q.ldy = Lr.out_dim;
          q.stats_out = x.at<double>(x.lo.stat_enc_layer[0]);
          q.M = rows; q.K = Lr.in_dim; q.Nout = Lr.out_dim;
          rc = mtmc::launch_few_l0(q, s);
        } else {
          mtmc::FewWaveParams q;
          q.A = x.at<float>(x.lo.Y[arg - 1]); q.lda = m->enc_node[arg - 1].out_dim;
          q.stats_in = x.at<double>(x.lo.stat_enc_layer[arg - 1]);
          q.gamma_in = m->enc_node[arg - 1].gamma; q.beta_in = m->enc_node[arg - 1].beta; q.count = (double)c->n_nodes;
          q.Wh = w_planes(x, arg); q.inv_w = w_inv(x, arg);
          q.bias = Lr.bias; q.Y = x.at<float>(x.lo.Y[arg]); q.ldy = Lr.out_dim;
          q.stats_out = x.at<double>(x.lo.stat_enc_layer[arg]);
          q.M = rows; q.K = Lr.in_dim; q.Nout = Lr.out_dim;
          q.drop_in = make_drop(x, m->dropout_enc); q.drop_stream = mtmc::kDropEncNode + arg - 1;
          if (x.enc2_rides && arg == m->n_enc_layers - 1) {
            const int64_t blocks = (c->n_edges + 255) / 256;
            q.pass_blocks = (int)(blocks > 2048 ? 2048 : blocks);
            q.pass_enc = enc_params(x); q.pass_attr = c->edge_attr; q.pass_edges = c->n_edges;
            q.pass_e_total = (double)c->n_edges_total; q.pass_stat = x.at<double>(x.lo.pub.stat_enc2_off);
          }
          rc = mtmc::launch_few_wave(q, s);
        }
        if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "encoder layer %d: few-row kernel refused the shape or the launch", arg);
        break;
      }
      if (arg == 0 && use_presplit0(x)) {
        if (phase == MTMC_PH_NODE_COMBINE) break;                // never split along K
        mtmc::SplitGemmParams q;
        q.Ah = x.at<_Float16>(x.lo.xh); q.inv_a = x.at<float>(x.lo.inv_a);
        q.Wh = w_planes(x, 0); q.inv_w = w_inv(x, 0);
        q.bias = Lr.bias; q.Y = x.at<float>(x.lo.Y[0]); q.ldy = Lr.out_dim;
        q.stats_out = x.at<double>(x.lo.stat_enc_layer[0]);
        q.amax_y = x.at<unsigned>(x.lo.amax) + (1 + MTMC_MAX_ENC_LAYERS) * mtmc::kAmaxRep;
        q.M = rows; q.K = Lr.in_dim; q.Nout = Lr.out_dim;
        if (x.pipe) {
          // Row panels: the side stream splits panel after panel (HBM-bound), the main stream multiplies panel i as soon as
          // its planes are there (matrix-bound): the split pass leaves the critical path except for the first panel.
          // Panel = whole rounds of workgroups of the tile height the whole matrix would get, so no round is paid twice.
          int64_t cuts[kMaxPanels + 1];
          const int np = l0_panels(rows, Lr.out_dim, cuts, &q.bm);
          q.M_rows = rows;
          if (hipEventRecord(x.pipe->fork, s) != hipSuccess || hipStreamWaitEvent(x.pipe->stream, x.pipe->fork, 0) != hipSuccess)
            return fail(MTMC_E_HIP, "encoder layer 0: fork onto the side stream failed");
          // from here on the side stream is forked off `s` (inside a capture: part of it): every way out joins it again
          auto join_and = [&](int code, const char* what) {
            if (hipEventRecord(x.pipe->ready[0], x.pipe->stream) == hipSuccess) (void)hipStreamWaitEvent(s, x.pipe->ready[0], 0);
            return fail(code, "%s", what);
          };
          for (int i = 0; i < np; ++i) {
            mtmc::launch_split_rows_range(c->x, c->x_row_stride, rows, Lr.in_dim, x.at<void>(x.lo.xh), x.at<float>(x.lo.inv_a),
                                          cuts[i], cuts[i + 1], x.pipe->stream);
            if (hipEventRecord(x.pipe->ready[i], x.pipe->stream) != hipSuccess) return join_and(MTMC_E_HIP, "hipEventRecord failed");
          }
          PanelTiming& pt = panel_timing();
          if (pt.on) pt.n = 0;
          for (int i = 0; i < np; ++i) {               // (the wait on the last panel's event also joins the side stream)
            if (hipStreamWaitEvent(s, x.pipe->ready[i], 0) != hipSuccess) return join_and(MTMC_E_HIP, "hipStreamWaitEvent failed");
            q.m_lo = cuts[i]; q.M = cuts[i + 1];
            if (pt.on) (void)hipEventRecord(pt.ev[2 * i], s);
            const int rc = mtmc::launch_gemm_presplit(q, s);
            if (pt.on) { (void)hipEventRecord(pt.ev[2 * i + 1], s); pt.n = i + 1; }
            if (rc != 0) return join_and(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "encoder layer 0: pre-split GEMM refused a panel");
          }
          break;
        }
        PanelTiming& pt = panel_timing();
        if (pt.on) { pt.n = 0; (void)hipEventRecord(pt.ev[0], s); }
        const int rc = mtmc::launch_gemm_presplit(q, s);
        if (pt.on) { (void)hipEventRecord(pt.ev[1], s); pt.n = 1; }
        if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "encoder layer 0: pre-split GEMM refused the shape or the launch");
        break;
      }
      if (use_staged(x, arg)) {
        if (phase == MTMC_PH_NODE_COMBINE) break;                // never split along K
        unsigned* amax = x.at<unsigned>(x.lo.amax);
        mtmc::StagedGemmParams q;
        q.A = x.at<float>(x.lo.Y[arg - 1]); q.lda = m->enc_node[arg - 1].out_dim;
        q.stats_in = x.at<double>(x.lo.stat_enc_layer[arg - 1]);
        q.gamma_in = m->enc_node[arg - 1].gamma; q.beta_in = m->enc_node[arg - 1].beta; q.count = (double)c->n_nodes;
        q.amax_a = amax + (1 + MTMC_MAX_ENC_LAYERS + (arg - 1)) * mtmc::kAmaxRep;
        q.Wh = w_planes(x, arg); q.inv_w = w_inv(x, arg);
        q.bias = Lr.bias; q.Y = x.at<float>(x.lo.Y[arg]); q.ldy = Lr.out_dim;
        q.stats_out = x.at<double>(x.lo.stat_enc_layer[arg]);
        q.amax_y = amax + (1 + MTMC_MAX_ENC_LAYERS + arg) * mtmc::kAmaxRep;
        q.M = rows; q.K = Lr.in_dim; q.Nout = Lr.out_dim;
        const int rc = mtmc::launch_gemm_staged(q, s);
        if (rc != 0) return fail(rc == MTMC_E_HIP ? MTMC_E_HIP : MTMC_E_ARG, "encoder layer %d: role-split GEMM refused the shape or the launch", arg);
        break;
      }
      mtmc::GemmParams g;
      if (arg == 0) {
        g.A = c->x; g.lda = c->x_row_stride; g.stats_in = nullptr; g.gamma_in = nullptr; g.beta_in = nullptr;
      } else {
        g.A = x.at<float>(x.lo.Y[arg - 1]); g.lda = m->enc_node[arg - 1].out_dim;
        g.stats_in = x.at<double>(x.lo.stat_enc_layer[arg - 1]);
        g.gamma_in = m->enc_node[arg - 1].gamma; g.beta_in = m->enc_node[arg - 1].beta;
      }
      g.W = Lr.weight; g.bias = Lr.bias; g.Y = x.at<float>(x.lo.Y[arg]); g.ldy = Lr.out_dim;
      g.count = (double)c->n_nodes; g.stats_out = x.at<double>(x.lo.stat_enc_layer[arg]);
      g.M = rows; g.K = Lr.in_dim; g.Nout = Lr.out_dim;
      g.drop_in = make_drop(x, m->dropout_enc); g.drop_stream = mtmc::kDropEncNode + arg - 1;
      {
        unsigned* amax = x.at<unsigned>(x.lo.amax);
        g.amax_a = arg == 0 ? amax : amax + (1 + MTMC_MAX_ENC_LAYERS + (arg - 1)) * mtmc::kAmaxRep;
        g.amax_w = x.at<unsigned>(x.lo.amax_w) + arg * mtmc::kAmaxRep;
        g.amax_y = amax + (1 + MTMC_MAX_ENC_LAYERS + arg) * mtmc::kAmaxRep;
      }
      if (x.enc2_rides && phase == MTMC_PH_NODE_ENC && arg == m->n_enc_layers - 1) {
        const int64_t blocks = (c->n_edges + 255) / 256;
        g.pass_blocks = (int)(blocks > 2048 ? 2048 : blocks);
        g.pass_enc = enc_params(x); g.pass_attr = c->edge_attr; g.pass_edges = c->n_edges;
        g.pass_e_total = (double)c->n_edges_total; g.pass_stat = x.at<double>(x.lo.pub.stat_enc2_off);
      }
      if (use_rows(x, arg)) {
        if (phase == MTMC_PH_NODE_COMBINE) break;                // never split along K
        g.slab = nullptr; g.split_k = 1;
        if (mtmc::launch_gemm_rows(g, s) != 0) return fail(MTMC_E_ARG, "encoder layer %d: row-streaming GEMM refused the shape", arg);
        break;
      }
      {  // the slab was sized for N rows; a shard with fewer rows may plan a larger split
        int sk_full, sk_here;
        mtmc::gemm_plan(c->n_nodes, g.K, g.Nout, &sk_full);
        mtmc::gemm_plan(rows, g.K, g.Nout, &sk_here);
        g.slab = (sk_here > 1 && (size_t)sk_here * rows <= (size_t)(sk_full > 1 ? sk_full : 0) * c->n_nodes)
                     ? x.at<float>(x.lo.slab) : nullptr;
        g.split_k = 1;
      }
      if (mtmc::launch_gemm_bn(g, s, phase == MTMC_PH_NODE_ENC ? 1 : 2) != MTMC_OK)
        return fail(MTMC_E_ARG, "encoder layer %d: unsupported GEMM shape", arg);
      break;
    }
    case MTMC_PH_NODE_H0: {
      const int last = m->n_enc_layers - 1;
      const int64_t rows = c->node_hi - c->node_lo;
      if (rows > 0)
        mtmc::launch_bn_relu_rows(x.at<float>(x.lo.Y[last]), MTMC_NODE_DIM, rows, MTMC_NODE_DIM,
                                  x.at<double>(x.lo.stat_enc_layer[last]), m->enc_node[last].gamma, m->enc_node[last].beta,
                                  (double)c->n_nodes, x.at<float>(x.lo.pub.h0_off) + (size_t)c->node_lo * MTMC_NODE_DIM,
                                  make_drop(x, m->dropout_enc), mtmc::kDropEncNode + last, c->node_lo, s);
      break;
    }
    case MTMC_PH_ROUND_PROJ: {
      if (arg < 0 || arg >= L) return fail(MTMC_E_ARG, "round %d out of range", arg);
      mtmc::NodeProjParams p;
      const int last = m->n_enc_layers - 1;
      p.y_last = (arg == 0 && fused_h0) ? x.at<float>(x.lo.Y[last]) : nullptr;
      p.y_stats = x.at<double>(x.lo.stat_enc_layer[last]); p.y_gamma = m->enc_node[last].gamma;
      p.y_beta = m->enc_node[last].beta; p.y_count = (double)c->n_nodes; p.h0_out = x.at<float>(x.lo.pub.h0_off);
      p.finalize_enc = arg == 0; p.enc = enc_params(x); p.e_total = (double)c->n_edges_total;
      p.h_src = round_h_src(x, arg);
      p.drop = make_drop(x, m->dropout_enc); p.drop_stream = mtmc::kDropEncNode + last;
      p.h0 = m->reattach_nodes ? x.at<float>(x.lo.pub.h0_off) : nullptr;
      p.deg = (m->agg == MTMC_AGG_MEAN && arg > 0) ? scale_deg(x) : nullptr;
      p.ue_w = m->upd_edge.weight; p.ue_ld = m->upd_edge.in_dim;
      p.un_w = m->upd_node.weight; p.un_ld = m->upd_node.in_dim;
      p.hn = (m->reattach_nodes ? 2 : 1) * MTMC_NODE_DIM;
      p.P = round_P(x, arg); p.Q = round_Q(x, arg);
      p.zero_buf = agg_target(x, arg);
      p.n_nodes = c->n_nodes;
      p.node_begin = proj_lo(c); p.node_end = proj_hi(c);
      p.edge_deg = x.at<int>(x.lo.pub.deg_off); p.un_b = m->upd_node.bias;
      p.z2_stats = mtmc::fold_node_stat(c->n_edges)
                       ? x.at<double>(x.lo.pub.stat_round_off) + (size_t)arg * mtmc::kRoundBlock + mtmc::kRoundZ2Off : nullptr;
      mtmc::launch_node_proj(p, s);
      break;
    }
    case MTMC_PH_ROUND_A:
    case MTMC_PH_ROUND_B:
    case MTMC_PH_ROUND_C: {
      if (arg < 0 || arg >= L) return fail(MTMC_E_ARG, "round %d out of range", arg);
      if (c->n_edges == 0) break;
      const mtmc::RoundParams p = round_params(x, arg);
      if (phase == MTMC_PH_ROUND_A) mtmc::launch_pass_a(p, s);
      else if (phase == MTMC_PH_ROUND_B) mtmc::launch_pass_b(p, s);
      else mtmc::launch_pass_c(p, s);
      break;
    }
    case MTMC_PH_ROUND_STAT: {     // few-edge lists: nothing (the statistics came out of MTMC_PH_ROUND_PROJ + MTMC_PH_ROUND_B)
      if (arg < 0 || arg >= L) return fail(MTMC_E_ARG, "round %d out of range", arg);
      if (mtmc::fold_node_stat(c->n_edges)) break;
      mtmc::NodeStatParams p;
      p.Q = round_Q(x, arg); p.deg = x.at<int>(x.lo.pub.deg_off); p.seg = x.at<double>(x.lo.pub.seg_off);
      p.un_w = m->upd_node.weight; p.un_b = m->upd_node.bias; p.un_ld = m->upd_node.in_dim;
      p.un_eoff = (m->reattach_nodes ? 2 : 1) * MTMC_NODE_DIM;
      p.stats = x.at<double>(x.lo.pub.stat_round_off) + (size_t)arg * mtmc::kRoundBlock;
      p.n_nodes = c->n_nodes;
      p.node_begin = proj_lo(c); p.node_end = proj_hi(c);
      mtmc::launch_node_stat(p, s);
      break;
    }
    case MTMC_PH_END: {
      const float* src = round_h_src(x, L);
      if (L == 0 || m->agg == MTMC_AGG_MEAN || x.lo.training)   // otherwise the last round aggregated into h_out
        mtmc::launch_h_final(src, scale_deg(x), (m->agg == MTMC_AGG_MEAN && L > 0) ? 1 : 0, c->n_nodes, c->h_out, s);
      if (L == 0 && c->n_edges > 0)
        mtmc::launch_classify_e0(enc_params(x), c->edge_attr, c->n_edges, (double)c->n_edges_total, m->cls.weight,
                                 m->cls.bias, m->cls.out_dim, c->logits, s);
      break;
    }
    default:
      return fail(MTMC_E_ARG, "unknown phase %d", phase);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MTMC_E_HIP, "kernel launch failed in phase %d: %s", phase, hipGetErrorString(e));
  return MTMC_OK;
}

}  // namespace mtmc_api
