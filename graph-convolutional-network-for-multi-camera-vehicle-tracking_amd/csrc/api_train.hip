// C-ABI entry point of the backward pass (include/mtmc_mpn.h: mtmc_mpn_backward).  Launch-only, like the forward.
// Order: rounds L-1..0 (node-update MLP, edge-update MLP + classifier, node projections), then the edge encoder,
// then the node encoder (BatchNorm backward + two fp32-MFMA GEMMs per layer on transposed operands).
#include "api_internal.h"
#include "train_kernels.h"

using namespace mtmc_api;

namespace {

#define HIP_OK(call)                                                                                   \
  do {                                                                                                 \
    if ((call) != hipSuccess) return fail(MTMC_E_HIP, "%s failed in mtmc_mpn_backward", #call);        \
  } while (0)

int check_grads(const mtmc_mpn_model* m, const mtmc_mpn_model* g) {
  if (!g) return fail(MTMC_E_ARG, "grads is NULL");
  if (g->struct_bytes != sizeof(mtmc_mpn_model)) return fail(MTMC_E_ARG, "grads.struct_bytes is %u, expected %zu", g->struct_bytes, sizeof(mtmc_mpn_model));
  for (int l = 0; l < m->n_enc_layers; ++l)
    if (!g->enc_node[l].weight || !g->enc_node[l].bias || !g->enc_node[l].gamma || !g->enc_node[l].beta)
      return fail(MTMC_E_ARG, "grads: NULL node-encoder gradient buffer");
  const mtmc_layer* ls[] = {&g->enc_edge[0], &g->enc_edge[1], &g->upd_edge, &g->upd_node};
  for (const mtmc_layer* l : ls)
    if (!l->weight || !l->bias || !l->gamma || !l->beta) return fail(MTMC_E_ARG, "grads: NULL gradient buffer");
  if (!g->cls.weight || !g->cls.bias) return fail(MTMC_E_ARG, "grads: NULL classifier gradient buffer");
  return MTMC_OK;
}

}  // namespace

static int backward_impl(const mtmc_mpn_model* model, const mtmc_mpn_call* call, const float* const* d_logits_steps,
                         const float* d_h, const mtmc_mpn_model* grads, void* grads_flat, size_t grads_flat_bytes,
                         float* d_x, float* d_edge_attr) {
  Ctx x;
  if (int rc = make_ctx(model, call, &x)) return rc;
  if (!call->training) return fail(MTMC_E_ARG, "mtmc_mpn_backward needs the call of a training-mode forward");
  if (int rc = check_grads(model, grads)) return rc;
  const mtmc_mpn_model* m = model;
  const int L = m->num_enc_steps;
  const int64_t N = call->n_nodes, E = call->n_edges;
  const int hn = (m->reattach_nodes ? 2 : 1) * MTMC_NODE_DIM;
  hipStream_t s = x.stream;
  const Layout& lo = x.lo;
  double* bst = x.at<double>(lo.bst);
  const size_t bst_block = (size_t)mtmc::kStatRep * kBwdStride;        // doubles per statistics block
  float* g_e[2] = {x.at<float>(lo.g_e[0]), x.at<float>(lo.g_e[1])};
  float* g_h[2] = {x.at<float>(lo.g_h[0]), x.at<float>(lo.g_h[1])};
  float* g_e0 = x.at<float>(lo.g_e0);
  float* g_h0 = x.at<float>(lo.g_h0);

  // gradient buffers are accumulated into with atomics: clear them first (the caller only provides storage).
  // One memset when the caller carved all of them out of one buffer (grads_flat), else one per tensor; the node
  // encoder's weight gradients are plain GEMM outputs and need none.
  // x^T and every W_l^T the node encoder's backward multiplies by depend on nothing the backward computes: they are made by the
  // backward's FIRST launch, beside the clearing of what it accumulates into
  float* tWl[MTMC_MAX_ENC_LAYERS] = {};
  float* tX = x.at<float>(lo.tX);
  const int64_t npad = (N + 31) / 32 * 32;
  mtmc::TransposeJobs tj;
  {
    mtmc::transpose_jobs_add(tj, call->x, N, m->enc_node[0].in_dim, call->x_row_stride, tX, npad);
    float* w = x.at<float>(lo.tW);
    for (int l = 0; l < m->n_enc_layers; ++l) {
      const mtmc_layer& Lr = m->enc_node[l];
      if (l > 0 || d_x) { tWl[l] = w; mtmc::transpose_jobs_add(tj, Lr.weight, Lr.out_dim, Lr.in_dim, Lr.in_dim, w, Lr.out_dim); }
      w += (size_t)Lr.in_dim * Lr.out_dim;
    }
  }
  bool transposed = false;
  auto zero = [&](float* p, size_t n) { return hipMemsetAsync(p, 0, n * sizeof(float), s); };
  bool ws_zeroed = false;
  if (grads_flat) {
    // one launch for the flat buffer's accumulated pieces (everything but the encoder's weight gradients, which must lie in it
    // in layer order, as mtmc_mpn_grad_layout lays them out) and the workspace's range; anything else: plain memsets
    mtmc::ZeroRanges z;
    char* at = static_cast<char*>(grads_flat);
    char* const end = at + grads_flat_bytes;
    bool ok = ((uintptr_t)at & 15) == 0 && (grads_flat_bytes & 15) == 0;
    for (int l = 0; ok && l < m->n_enc_layers; ++l) {
      char* w = reinterpret_cast<char*>(const_cast<float*>(grads->enc_node[l].weight));
      const size_t wb = (size_t)m->enc_node[l].in_dim * m->enc_node[l].out_dim * sizeof(float);
      ok = w >= at && w + wb <= end && ((uintptr_t)w & 15) == 0 && (wb & 15) == 0;
      if (ok && w > at) { z.r[z.n].p = reinterpret_cast<uint4*>(at); z.r[z.n].n16 = (size_t)(w - at) / 16; ++z.n; }
      at = w + wb;
    }
    if (ok) {
      if (end > at) { z.r[z.n].p = reinterpret_cast<uint4*>(at); z.r[z.n].n16 = (size_t)(end - at) / 16; ++z.n; }
      ok = ((uintptr_t)(x.ws + lo.bwd_zero) & 15) == 0 && ((lo.bwd_zero_end - lo.bwd_zero) & 15) == 0;
    }
    if (ok) {
      z.r[z.n].p = reinterpret_cast<uint4*>(x.ws + lo.bwd_zero); z.r[z.n].n16 = (lo.bwd_zero_end - lo.bwd_zero) / 16; ++z.n;
      mtmc::launch_bwd_begin(z, tj, s);
      ws_zeroed = transposed = true;
    } else {
      HIP_OK(hipMemsetAsync(grads_flat, 0, grads_flat_bytes, s));
    }
  } else {
    for (int l = 0; l < m->n_enc_layers; ++l) {
      const mtmc_layer& g = grads->enc_node[l];
      const size_t o = m->enc_node[l].out_dim;
      HIP_OK(zero(const_cast<float*>(g.bias), o));
      HIP_OK(zero(const_cast<float*>(g.gamma), o)); HIP_OK(zero(const_cast<float*>(g.beta), o));
    }
    const mtmc_layer* gs[] = {&grads->enc_edge[0], &grads->enc_edge[1], &grads->upd_edge, &grads->upd_node, &grads->cls};
    const mtmc_layer* ms[] = {&m->enc_edge[0], &m->enc_edge[1], &m->upd_edge, &m->upd_node, &m->cls};
    for (int i = 0; i < 5; ++i) {
      const size_t o = ms[i]->out_dim, in = ms[i]->in_dim;
      HIP_OK(zero(const_cast<float*>(gs[i]->weight), o * in)); HIP_OK(zero(const_cast<float*>(gs[i]->bias), o));
      if (i < 4) { HIP_OK(zero(const_cast<float*>(gs[i]->gamma), o)); HIP_OK(zero(const_cast<float*>(gs[i]->beta), o)); }
    }
  }
  // the workspace side: statistics blocks, per-round dP/dQ, dh0, de0, the first de / dh buffers -- one range
  if (!ws_zeroed) HIP_OK(hipMemsetAsync(x.ws + lo.bwd_zero, 0, lo.bwd_zero_end - lo.bwd_zero, s));
  if (!transposed) mtmc::launch_transpose_multi(tj, s);
  int cur = 0, cur_e = 0;
  if (d_h) HIP_OK(hipMemcpyAsync(L > 0 ? g_h[cur] : g_h0, d_h, (size_t)N * 32 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (L == 0 && d_logits_steps && d_logits_steps[0] && E > 0)       // no rounds: classifier on the encoded edges
    mtmc::launch_bwd_classify_e0(enc_params(x), call->edge_attr, E, (double)E, m->cls.weight, m->cls.out_dim,
                                 d_logits_steps[0], g_e0, const_cast<float*>(grads->cls.weight),
                                 const_cast<float*>(grads->cls.bias), s);

  int first_cls = L - m->num_class_steps + 1;
  if (first_cls < 1) first_cls = 1;

  for (int r = L - 1; r >= 0; --r) {
    // P, Q of this round are on the tape (Layout::P_tr / Q_tr); what bwd_node_proj needs of the forward's node_proj:
    const float* h_src_r = round_h_src(x, r);
    const float* h0_r = m->reattach_nodes ? x.at<float>(lo.pub.h0_off) : nullptr;
    const int* deg_r = (m->agg == MTMC_AGG_MEAN && r > 0) ? x.at<int>(lo.pub.deg_off) : nullptr;

    mtmc::BwdRoundParams bp;
    bp.f = round_params(x, r);
    bp.g_h = g_h[cur]; bp.h_agg = x.at<float>(lo.h_tr[r]); bp.deg = x.at<int>(lo.pub.deg_off);
    bp.arg = x.at<int>(lo.g_arg);
    const int step = r + 1;
    bp.d_logits = (d_logits_steps && step >= first_cls) ? d_logits_steps[step - first_cls] : nullptr;
    bp.g_de2 = x.at<float>(lo.g_de2);
    bp.g_Q = x.at<float>(lo.g_Q) + (size_t)r * N * 32; bp.g_P = x.at<float>(lo.g_P) + (size_t)r * mtmc::kGradRep * N * 8;
    bp.g_e = g_e[cur_e]; bp.g_e_prev = g_e[cur_e ^ 1]; bp.g_e0 = g_e0; bp.bst = bst + (size_t)(2 * r) * bst_block;
    bp.gr_un_w = const_cast<float*>(grads->upd_node.weight); bp.gr_un_b = const_cast<float*>(grads->upd_node.bias);
    bp.gr_un_g = const_cast<float*>(grads->upd_node.gamma); bp.gr_un_bt = const_cast<float*>(grads->upd_node.beta);
    bp.gr_ue_w = const_cast<float*>(grads->upd_edge.weight); bp.gr_ue_b = const_cast<float*>(grads->upd_edge.bias);
    bp.gr_ue_g = const_cast<float*>(grads->upd_edge.gamma); bp.gr_ue_bt = const_cast<float*>(grads->upd_edge.beta);
    bp.gr_cls_w = const_cast<float*>(grads->cls.weight); bp.gr_cls_b = const_cast<float*>(grads->cls.bias);
    bp.gacc = x.at<float>(lo.gacc);

    if (m->agg == MTMC_AGG_MAX) {
      HIP_OK(hipMemsetAsync(bp.arg, 0x7f, (size_t)N * 32 * sizeof(int32_t), s));
      mtmc::launch_bwd_node_upd(bp, 2, s);
    }
    mtmc::launch_bwd_node_upd(bp, 0, s);
    mtmc::launch_bwd_node_upd(bp, 1, s);
    bp.bst = bst + (size_t)(2 * r + 1) * bst_block;
    mtmc::launch_bwd_edge_upd(bp, 0, s);
    mtmc::launch_bwd_edge_upd(bp, 1, s);

    mtmc::BwdProjParams pp;
    pp.g_P = bp.g_P; pp.g_Q = bp.g_Q; pp.h_src = h_src_r; pp.h0 = h0_r; pp.deg = deg_r;
    pp.ue_w = m->upd_edge.weight; pp.ue_ld = m->upd_edge.in_dim; pp.un_w = m->upd_node.weight; pp.un_ld = m->upd_node.in_dim;
    pp.hn = hn;
    pp.g_h_prev = g_h[cur ^ 1]; pp.g_h0 = g_h0; pp.src_is_h0 = r == 0;
    pp.gr_ue_w = bp.gr_ue_w; pp.gr_un_w = bp.gr_un_w; pp.n_nodes = N;
    mtmc::launch_bwd_node_proj(pp, s);
    cur ^= 1;
    cur_e ^= 1;
  }

  // ---- edge encoder -------------------------------------------------------------------------------
  {
    mtmc::BwdEncParams ep;
    ep.enc = enc_params(x); ep.attr = call->edge_attr; ep.n_edges = E; ep.e_total = (double)E; ep.g_e0 = g_e0;
    ep.bst = bst + (size_t)(2 * L) * bst_block; ep.d_attr = d_edge_attr;
    ep.gacc = x.at<float>(lo.gacc);
    ep.gr_w1 = const_cast<float*>(grads->enc_edge[0].weight); ep.gr_b1 = const_cast<float*>(grads->enc_edge[0].bias);
    ep.gr_g1 = const_cast<float*>(grads->enc_edge[0].gamma); ep.gr_bt1 = const_cast<float*>(grads->enc_edge[0].beta);
    ep.gr_w2 = const_cast<float*>(grads->enc_edge[1].weight); ep.gr_b2 = const_cast<float*>(grads->enc_edge[1].bias);
    ep.gr_g2 = const_cast<float*>(grads->enc_edge[1].gamma); ep.gr_bt2 = const_cast<float*>(grads->enc_edge[1].beta);
    for (int pass = 0; pass < 3; ++pass) mtmc::launch_bwd_edge_enc(ep, pass, s);
  }

  // ---- node encoder -------------------------------------------------------------------------------
  {
    float* gA = x.at<float>(lo.gA);
    float* gB = x.at<float>(lo.gB);
    float* tA = x.at<float>(lo.tA);
    float* tB = x.at<float>(lo.tB);
    float* zeros = x.at<float>(lo.zeros);
    const mtmc::Drop nodrop = {0, 0, 1.f, 0};
    for (int l = m->n_enc_layers - 1; l >= 0; --l) {
      const mtmc_layer& Lr = m->enc_node[l];
      const int d = Lr.out_dim, in = Lr.in_dim;
      // dh0 is the last layer's dA: bn_bwd turns it into dY in place, where the edge rounds' backward left it
      float* dY = (l == m->n_enc_layers - 1) ? g_h0 : gA;
      size_t sb_off = 0;
      for (int j = 0; j < l; ++j) sb_off += 2 * (size_t)m->enc_node[j].out_dim;
      double* sb = x.at<double>(lo.bst_n) + sb_off;
      mtmc::BnBwdParams bb;
      bb.Y = x.at<float>(lo.Y[l]); bb.dA = dY; bb.rows = N; bb.dim = d;
      bb.stats_fwd = x.at<double>(lo.stat_enc_layer[l]); bb.stats_bwd = sb; bb.count = (double)N;
      bb.gamma = Lr.gamma; bb.beta = Lr.beta; bb.drop = make_drop(x, m->dropout_enc); bb.drop_stream = mtmc::kDropEncNode + l;
      bb.gr_gamma = const_cast<float*>(grads->enc_node[l].gamma); bb.gr_beta = const_cast<float*>(grads->enc_node[l].beta);
      bb.gr_bias = const_cast<float*>(grads->enc_node[l].bias);
      // |.|max of the three GEMM operands of this layer: dY_l (written by bn_bwd's apply pass), a_{l-1} (x: the forward's
      // value; else from the recomputation below) and W_l (the forward's) -> the fp16 three-product kernel applies
      unsigned* amax_fwd = x.at<unsigned>(lo.amax);
      unsigned* amax_dy = x.at<unsigned>(lo.amax_bwd) + (size_t)l * mtmc::kAmaxRep;
      unsigned* amax_act = x.at<unsigned>(lo.amax_bwd) + (size_t)(MTMC_MAX_ENC_LAYERS + l) * mtmc::kAmaxRep;
      bb.amax_out = amax_dy;
      bb.dT = tA; bb.ldt = npad;                                       // dY_l^T comes out of the apply pass directly
      // the layer's input activation a_{l-1}: x itself, or relu(bn(Y_{l-1})) with its dropout mask -- recomputed (with its
      // transpose and its |.|max) by the z = 1 workgroups of the apply launch: nothing of it depends on dY_l
      const float* aT = tX;                                            // a_{l-1}^T [in][npad]
      const bool ride = l > 0 && mtmc::bn_bwd_carries_rows_job(N, npad);
      if (l > 0) {
        const mtmc_layer& Pv = m->enc_node[l - 1];
        bb.rc = {x.at<float>(lo.Y[l - 1]), Pv.out_dim, N, Pv.out_dim, x.at<double>(lo.stat_enc_layer[l - 1]), Pv.gamma, Pv.beta,
                 (double)N, gB, make_drop(x, m->dropout_enc), mtmc::kDropEncNode + l - 1, 0, amax_act, tB, npad};
        bb.rc_on = ride ? 1 : 0;
        aT = tB;
      }
      mtmc::launch_bn_bwd(bb, 0, s);
      mtmc::launch_bn_bwd(bb, 1, s);                                   // dY now holds dY_l (+ a_{l-1}, a_{l-1}^T)
      if (l > 0 && !ride) {
        const mtmc_layer& Pv = m->enc_node[l - 1];
        mtmc::launch_bn_relu_rows(x.at<float>(lo.Y[l - 1]), Pv.out_dim, N, Pv.out_dim, x.at<double>(lo.stat_enc_layer[l - 1]),
                                  Pv.gamma, Pv.beta, (double)N, gB, make_drop(x, m->dropout_enc), mtmc::kDropEncNode + l - 1, 0, s,
                                  amax_act, tB, npad);
      }
      // dW_l [d][in] = dY^T . a  -> NT GEMM on the transposes (reduction over the node rows, padded to 32)
      mtmc::GemmParams g;
      g.A = tA; g.lda = npad; g.W = aT; g.bias = zeros; g.Y = const_cast<float*>(grads->enc_node[l].weight); g.ldy = in;
      g.stats_in = nullptr; g.gamma_in = nullptr; g.beta_in = nullptr; g.count = 1; g.stats_out = nullptr;
      g.M = d; g.K = (int)npad; g.Nout = in; g.drop_in = nodrop; g.drop_stream = 0; g.slab = nullptr; g.split_k = 1;
      g.amax_a = amax_dy; g.amax_w = l > 0 ? amax_act : amax_fwd; g.amax_y = nullptr;
      if (mtmc::launch_gemm_bn(g, s) != MTMC_OK) return fail(MTMC_E_ARG, "backward: weight-gradient GEMM shape");
      // dA_{l-1} [N][in] = dY . W_l  -> NT GEMM against W^T
      if (l > 0 || d_x) {
        g.A = dY; g.lda = d; g.W = tWl[l]; g.Y = l > 0 ? gB : d_x; g.ldy = in; g.M = N; g.K = d; g.Nout = in;    // W_l^T [in][d]
        g.amax_w = x.at<unsigned>(lo.amax_w) + (size_t)l * mtmc::kAmaxRep;
        if (mtmc::launch_gemm_bn(g, s) != MTMC_OK) return fail(MTMC_E_ARG, "backward: input-gradient GEMM shape");
        std::swap(gA, gB);
      }
    }
  }
  {  // the replicated small-gradient sums of the edge kernels -> the caller's tensors
    static_assert(mtmc::kGradRep == 16 && mtmc::kGaccN == 256, "Layout::gacc is sized for 16 x 256 floats");
    mtmc::GradFoldParams fp;
    fp.gacc = x.at<float>(lo.gacc);
    fp.gr_un_w = const_cast<float*>(grads->upd_node.weight); fp.gr_un_b = const_cast<float*>(grads->upd_node.bias);
    fp.un_ld = m->upd_node.in_dim; fp.un_eoff = hn;
    fp.gr_ue_w = const_cast<float*>(grads->upd_edge.weight); fp.gr_ue_b = const_cast<float*>(grads->upd_edge.bias);
    fp.ue_ld = m->upd_edge.in_dim; fp.ue_eoff = 2 * hn; fp.nin = m->reattach_edges ? 8 : 4;
    fp.gr_cls_w = const_cast<float*>(grads->cls.weight); fp.gr_cls_b = const_cast<float*>(grads->cls.bias);
    fp.n_classes = m->cls.out_dim;
    fp.gr_w1 = const_cast<float*>(grads->enc_edge[0].weight); fp.gr_b1 = const_cast<float*>(grads->enc_edge[0].bias);
    fp.gr_w2 = const_cast<float*>(grads->enc_edge[1].weight); fp.gr_b2 = const_cast<float*>(grads->enc_edge[1].bias);
    fp.fe = m->enc_edge[0].in_dim;
    mtmc::launch_grad_fold(fp, s);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MTMC_E_HIP, "kernel launch failed in backward: %s", hipGetErrorString(e));
  return MTMC_OK;
}

extern "C" int32_t mtmc_mpn_backward(const mtmc_mpn_model* model, const mtmc_mpn_call* call, const float* d_logits,
                                     const float* d_h, const mtmc_mpn_model* grads, float* d_x, float* d_edge_attr) {
  const float* steps[64];
  int n_out = 0;
  if (int rc = check_model(model)) return rc;          // (incl. the struct-size guards, before any field is trusted)
  if (int rc = check_call_size(call)) return rc;
  if (d_logits) {
    n_out = model->num_enc_steps > 0 ? std::min(model->num_class_steps, model->num_enc_steps) : 1;
    if (n_out > 64) return fail(MTMC_E_ARG, "more than 64 classified steps");
    for (int i = 0; i < n_out; ++i) steps[i] = d_logits + (size_t)i * call->n_edges * model->cls.out_dim;
  }
  return backward_impl(model, call, d_logits ? steps : nullptr, d_h, grads, nullptr, 0, d_x, d_edge_attr);
}

extern "C" int32_t mtmc_mpn_backward_steps(const mtmc_mpn_model* model, const mtmc_mpn_call* call,
                                           const float* const* d_logits_steps, const float* d_h,
                                           const mtmc_mpn_model* grads, void* grads_flat, size_t grads_flat_bytes,
                                           float* d_x, float* d_edge_attr) {
  return backward_impl(model, call, d_logits_steps, d_h, grads, grads_flat, grads_flat_bytes, d_x, d_edge_attr);
}

// The same with the gradient carving done here: `flat` (fp32, flat_floats long) receives every parameter gradient in
// struct order -- node encoder layers, edge encoder, edge update, node update, classifier; weight, bias, then gamma, beta
// where the layer has a BatchNorm -- each piece starting on a 64-float (256-byte) boundary.  The host makes ONE allocation
// and views it with the same rule (mtmc_mpn_grad_layout: offsets in floats, returns the total), instead of filling a
// second 34-pointer struct per call.
extern "C" int64_t mtmc_mpn_grad_layout(const mtmc_mpn_model* m, int64_t* offsets, int32_t max_offsets) {
  if (!m || m->struct_bytes != sizeof(mtmc_mpn_model) || m->n_enc_layers < 1 || m->n_enc_layers > MTMC_MAX_ENC_LAYERS)
    return 0;                                                                           // (0 = no layout: bad model)
  int64_t total = 0;
  int n = 0;
  auto piece = [&](int64_t numel) {
    if (offsets && n < max_offsets) offsets[n] = total;
    ++n;
    total += (numel + 63) / 64 * 64;
  };
  auto layer = [&](const mtmc_layer& l, bool bn) {
    piece((int64_t)l.out_dim * l.in_dim);
    piece(l.out_dim);
    if (bn) { piece(l.out_dim); piece(l.out_dim); }
  };
  for (int l = 0; l < m->n_enc_layers; ++l) layer(m->enc_node[l], true);
  layer(m->enc_edge[0], true); layer(m->enc_edge[1], true);
  layer(m->upd_edge, true); layer(m->upd_node, true);
  layer(m->cls, false);
  return total;
}

extern "C" int32_t mtmc_mpn_backward_flat(const mtmc_mpn_model* model, const mtmc_mpn_call* call,
                                          const float* const* d_logits_steps, const float* d_h, float* flat,
                                          int64_t flat_floats, float* d_x, float* d_edge_attr) {
  if (!model || !flat) return fail(MTMC_E_ARG, "mtmc_mpn_backward_flat: NULL model or gradient buffer");
  if (int rc = check_model(model)) return rc;          // before anything indexes enc_node[] / off[] by n_enc_layers
  int64_t off[4 * (MTMC_MAX_ENC_LAYERS + 4) + 2];
  const int64_t need = mtmc_mpn_grad_layout(model, off, (int32_t)(sizeof(off) / sizeof(off[0])));
  if (flat_floats < need) return fail(MTMC_E_ARG, "mtmc_mpn_backward_flat: gradient buffer too small");
  mtmc_mpn_model g = *model;
  int n = 0;
  auto layer = [&](mtmc_layer& l, bool bn) {
    l.weight = flat + off[n++];
    l.bias = flat + off[n++];
    l.gamma = bn ? flat + off[n++] : nullptr;
    l.beta = bn ? flat + off[n++] : nullptr;
  };
  for (int l = 0; l < g.n_enc_layers; ++l) layer(g.enc_node[l], true);
  layer(g.enc_edge[0], true); layer(g.enc_edge[1], true);
  layer(g.upd_edge, true); layer(g.upd_node, true);
  layer(g.cls, false);
  return backward_impl(model, call, d_logits_steps, d_h, &g, flat, (size_t)need * sizeof(float), d_x, d_edge_attr);
}

