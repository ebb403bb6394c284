// Launch interface of the backward kernels (train_kernels.hip).
#pragma once
#include "kernels.h"
#include "rows_body.h"

namespace mtmc {

constexpr int kGradRep = 16;   // replicas of the dP accumulator (see bwd_edge_upd_kernel)

// Sums that EVERY workgroup of an edge kernel adds to the same few gradient words (biases, the 4-wide edge columns
// of the update weights, the classifier, the edge encoder): same-address float atomics serialise in L2 at ~130 ns
// each, i.e. ~90 us behind a 676-workgroup launch.  They go to one of kGradRep replica rows of a workspace scratch
// instead (row = workgroup % kGradRep, zeroed with the rest of the backward scratch) and grad_fold_kernel adds the
// rows into the caller's gradient tensors once, at the end of the backward.
constexpr int kGaUnB = 0, kGaUnW = 32, kGaClsW = 160, kGaClsB = kGaClsW + 4 * MTMC_MAX_CLASSES,
              kGaUeB = kGaClsB + MTMC_MAX_CLASSES, kGaUeW = kGaUeB + 4, kGaW2 = kGaUeW + 32, kGaB2 = kGaW2 + 16,
              kGaW1 = kGaB2 + 4, kGaB1 = kGaW1 + 8, kGaccN = 256;
static_assert(kGaB1 + 4 <= kGaccN, "gradient scratch row too short");

constexpr int kBwdStrideD = 256;   // doubles per replica of the backward statistics scratch (api_internal.h: kBwdStride)

struct BwdRoundParams {
  RoundParams f;             // the forward parameters of this round (tape pointers, weights, statistics, dropout)
  const float* g_h;          // [N][32] gradient wrt the round's aggregated node state (as consumed: mean-scaled)
  const float* h_agg;        // [N][32] the aggregated values (max aggregation routes the gradient to the arg max)
  const int* deg;
  int* arg;                  // [N][32] max aggregation: edge index of the arg max per (node, channel)
  const float* d_logits;     // [E][C] gradient of this round's logits, or nullptr
  float* g_de2;              // [E][4] scratch: A^T . (gradient wrt the node-update pre-activation)
  float* g_Q; float* g_P;    // [N][32], [kGradRep]{dPr [N][4], dPc [4][N]} (zeroed by the host before the round)
  float* g_e;                // [E][4] in: gradient wrt e_r from later rounds; becomes g1 (mode 0 of the edge kernel)
  float* g_e_prev;           // [E][4] out: gradient wrt e_{r-1}
  float* g_e0;               // [E][4] accumulated gradient wrt the encoded edges
  double* bst;               // backward statistics scratch (zeroed by the host before each statistics pass)
  float* gr_un_w; float* gr_un_b; float* gr_un_g; float* gr_un_bt;
  float* gr_ue_w; float* gr_ue_b; float* gr_ue_g; float* gr_ue_bt;
  float* gr_cls_w; float* gr_cls_b;
  float* gacc;               // [kGradRep][kGaccN] replicated small-gradient sums (see above)
};

struct GradFoldParams {
  const float* gacc;
  float* gr_un_w; float* gr_un_b; int un_ld, un_eoff;
  float* gr_ue_w; float* gr_ue_b; int ue_ld, ue_eoff, nin;
  float* gr_cls_w; float* gr_cls_b; int n_classes;
  float* gr_w1; float* gr_b1; float* gr_w2; float* gr_b2; int fe;
};

struct BwdProjParams {
  const float* g_P; const float* g_Q;
  const float* h_src; const float* h0; const int* deg;
  const float* ue_w; int ue_ld; const float* un_w; int un_ld; int hn;
  float* g_h_prev; float* g_h0; int src_is_h0;
  float* gr_ue_w; float* gr_un_w;
  int64_t n_nodes;
};

struct BwdEncParams {
  EdgeEncParams enc; const float* attr; int64_t n_edges; double e_total;
  const float* g_e0; double* bst; float* d_attr;
  float* gacc;               // [kGradRep][kGaccN]: w1, b1, w2, b2 sums
  float* gr_w1; float* gr_b1; float* gr_g1; float* gr_bt1;
  float* gr_w2; float* gr_b2; float* gr_g2; float* gr_bt2;
};

struct BnBwdParams {
  const float* Y; float* dA; int64_t rows; int dim;
  const double* stats_fwd; double* stats_bwd; double count;
  const float* gamma; const float* beta; Drop drop; unsigned drop_stream;
  float* gr_gamma; float* gr_beta; float* gr_bias;
  unsigned* amax_out;        // u32[kAmaxRep] |dY|max (mode 1; atomicMax on the bit patterns), or nullptr
  float* dT; int64_t ldt;    // mode 1: dY^T [dim][ldt] as well (rows..ldt zero-filled): the weight-gradient GEMM's operand
  RowsTJob rc; int rc_on = 0;  // mode 1: the recomputation of the layer's INPUT activation (+ its transpose) rides as the z = 1
                               // workgroups of the launch (independent of this layer's dY); same rows, 16 per workgroup
};

void launch_bwd_node_upd(const BwdRoundParams& p, int mode, hipStream_t s);
void launch_grad_fold(const GradFoldParams& p, hipStream_t s);
void launch_bwd_edge_upd(const BwdRoundParams& p, int mode, hipStream_t s);
void launch_bwd_node_proj(const BwdProjParams& p, hipStream_t s);
void launch_bwd_edge_enc(const BwdEncParams& p, int pass, hipStream_t s);
void launch_bwd_classify_e0(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total, const float* cls_w,
                            int n_classes, const float* d_logits, float* g_e0, float* gr_cls_w, float* gr_cls_b,
                            hipStream_t s);
void launch_bn_bwd(const BnBwdParams& p, int mode, hipStream_t s);
bool bn_bwd_carries_rows_job(int64_t rows, int64_t ldt);      // the shapes bn_relu_rows_t_body takes
void launch_transpose_pad(const float* src, int64_t rows, int cols, int64_t ld_src, float* dst, int64_t rows_pad,
                          hipStream_t s);
// what the backward accumulates into, cleared by ONE launch: the gaps between the node encoder's weight gradients in the caller's
// flat gradient buffer (those are plain GEMM outputs) and the workspace's zero range (two memsets, 10.8 MB of them needless, before)
struct ZeroRanges { int n = 0; struct { uint4* p; size_t n16; } r[MTMC_MAX_ENC_LAYERS + 3]; };
void launch_zero_ranges(const ZeroRanges& z, hipStream_t s);
// several padded transposes in one launch (x^T and every W_l^T of the node encoder's backward)
struct TransposeJob { const float* src; float* dst; int64_t rows, ld_src, rows_pad; int cols; unsigned blocks_r, first_block; };
struct TransposeJobs { int n = 0; unsigned n_blocks = 0; TransposeJob job[MTMC_MAX_ENC_LAYERS + 1]; };
void transpose_jobs_add(TransposeJobs& p, const float* src, int64_t rows, int cols, int64_t ld_src, float* dst, int64_t rows_pad);
void launch_transpose_multi(const TransposeJobs& p, hipStream_t s);
void launch_bwd_begin(const ZeroRanges& z, const TransposeJobs& t, hipStream_t s);    // both in one launch (the backward's first)

}  // namespace mtmc
