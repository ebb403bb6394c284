// Internal launch interface between the C-ABI layer (api.hip) and the kernel translation units.
#pragma once
#include "../../include/mtmc_mpn.h"
#include "common.h"

namespace mtmc {

// Optional A/B switches from the environment, read ONCE per process (knobs.hip) -- the launch paths
// only look at this struct.
struct Knobs {
  bool pass_c_walk;          // MTMC_PASS_C_WALK: pass C on the half-wave walk everywhere
  int64_t pass_c_small_min;  // MTMC_PASS_C_SMALL_MIN: fewest edges for the matrix-core pass C on few-edge lists
  int staged_xr;             // MTMC_STAGED_XR: rows (16..64) of the shorter consumer waves of the role-split GEMM (0 = per launch)
  int presplit_rows;         // MTMC_PRESPLIT_ROWS: tile height of the layer-0 GEMM (0 = chosen per launch)
  bool pass_c_general;       // MTMC_PASS_C_GENERAL: many-edge sorted lists on the any-order matrix-core kernel (A/B)
  int pass_c_span;           // MTMC_PASS_C_SPAN: 64-edge chunks per wave span (0 = default)
  int pass_c_blocks;         // MTMC_PASS_C_BLOCKS: resident grid of the matrix-core pass C
  bool gemm_fp32;            // MTMC_GEMM_FP32: exact-fp32 MFMA encoder everywhere
  bool gemm_no_f16;          // MTMC_GEMM_NO_F16: bf16x6 (large) / exact fp32 (few rows) instead of the fp16 split kernels
  bool gemm_no_presplit;     // MTMC_GEMM_NO_PRESPLIT: layer 0 of many-row graphs on the in-loop kernel
  bool gemm_no_staged;       // MTMC_GEMM_NO_STAGED: layers >= 1 of many-row graphs on the in-loop kernel
  bool no_col_blocks;        // MTMC_NO_COL_BLOCKS: pass A in edge order everywhere (A/B of the column-blocked traversal)
  int col_blocks;            // MTMC_COL_BLOCKS: number of column blocks (8..64, multiple of 8; 0 = from the table size)
  int l0_pipeline;           // MTMC_L0_PIPELINE: layer 0 of many-row graphs in row panels, the operand split of panel i+1 on a
                             // side stream beside panel i's GEMM.  0 = off (one split pass, one GEMM), 1 = panels of
                             // 1, 2, 4, 8, 8, ... rounds of workgroups (default), 2 / 3 = uniform panels of 1 / 2 rounds
  bool gemm_no_few;          // MTMC_GEMM_NO_FEW: few-row graphs on the split-K in-loop kernels + combine (rounds 1-4; A/B)
  int few_rows_max;          // MTMC_FEW_ROWS_MAX: most node rows a call may encode on the few-row kernels (gemm_few.hip)
  int few_l0_map;            // MTMC_FEW_L0_MAP: tile -> XCD binding of few_l0_kernel (0: column groups, 1: 4 x 8 tiles per XCD, 2: none)
  int few_wave_rb;           // MTMC_FEW_WAVE_RB: 16-row blocks per workgroup of few_wave_kernel (0: chosen per launch)
};
const Knobs& knobs();

// |.|max of a [rows][cols] fp32 matrix (row stride ld), accumulated with atomicMax on the bit pattern: scales of the
// fp16 two-piece node-encoder GEMM.  Done by extra workgroups of prep_kernel, i.e. without a launch of its own.
constexpr int kAmaxRep = 16;   // replicas of every |.|max word: same-address atomics serialise in L2
// kind 1 (kJobSplit): the operand split of split_body.h instead -- rows of `ptr` -> fp16 planes + row scales, 8 rows per
// workgroup; fp != nullptr: a chunk is only split again when its 64-bit fingerprint differs from the stored one (the
// content-verified weight-plane cache)
enum { kJobAmax = 0, kJobSplit = 1 };
struct AmaxJob { const float* ptr; int64_t rows; int cols; int64_t ld; unsigned* out; int block0, n_blocks;   // block range: set by launch_prep
                 int kind; _Float16* planes; float* inv; unsigned long long* fp; };

// The edge part of prep_kernel: int64 strided edge_index -> int32 row / col, out-degree, sortedness flags, moments of edge_attr
struct PrepEdge {
  int n_edge_blocks = 0;              // workgroups that share the edge list (set by the launcher)
  const int64_t* row; const int64_t* col; int64_t idx_stride;
  const float* attr; int fe;
  int64_t n_edges; int64_t n_nodes;
  int* row32; int* col32; int* deg; int* flags; int* row_start;
  double* stat_attr;
};
struct PrepParams : PrepEdge {
  int n_jobs;                         // grid = passenger blocks working through jobs[] (first) + n_edge_blocks
  int n_pass_blocks;                  // set by launch_prep
  AmaxJob jobs[2 * MTMC_MAX_ENC_LAYERS + 2];
};

// Everything one message-passing round needs (passes A, B, C).
struct RoundParams {
  const int* row32; const int* col32; const float* attr;
  const float* e_prev;       // [E][4] edge state of the previous round (unused in the first round)
  float* e_buf;              // [E][4] this round: z1 written by pass A
  float* e_out;              // [E][4] e' written by pass B, read by pass C (eval: == e_buf, in place; training: kept apart)
  Drop drop_e, drop_n; unsigned drop_stream;   // training: dropout of the edge / node update MLPs
  const float* P;            // [2][N][4]: Pr rows, then Pc rows (the gathered half compact)
  const float* Q;            // [N][32]
  double* seg;               // [N][4] per-node segment sums of e' (many-edge lists only: fold_z2 == 0)
  int fold_z2;               // 1: pass B adds the edge part of the node-update statistics itself (few-edge lists)
  const float* ue_w; const float* ue_b; const float* ue_g; const float* ue_bt; int ue_ld; int ue_eoff;
  const float* un_w; const float* un_b; const float* un_g; const float* un_bt; int un_ld; int un_eoff;
  const float* cls_w; const float* cls_b; int n_classes;
  double* stats;             // this round's kRoundStride doubles
  float* h_acc;              // [N][32] aggregation target (pre-zeroed)
  float* logits;             // this round's [E][C] output or nullptr
  int64_t n_edges; double e_total;
  int first_round; int reattach_edges; int agg;
  // eval mode: e' = relu(bn(z1)) is never written -- pass C and the next round's pass A recompute it from z1 (4 FMAs)
  // with the round's z1 statistics; pass B only reduces it.  16 B/edge/round less traffic.  prev_stats: round r-1's block
  int lazy_e; const double* prev_stats;
  int mfma_c;                // set by launch_pass_c: the matrix-core pass C takes row-sorted lists
  int stream_z1;             // set by launch_pass_a: z1 is stored non-temporally (lists whose z1 outgrows the Infinity Cache)
  int det_len;               // set by launch_pass_c: edges per carry chunk of the deterministic sums (32, or a span of the sorted kernel)
  int det; const int* flags; const int* deg; const int* row_start; float* carry; int64_t n_nodes;   // deterministic sums
  // edges per SOURCE ROW of this call's edges: local edges / owned rows for a row-complete shard, E_total / N otherwise
  // (a shard's local edge count over the GLOBAL node count would send 8 ranks of config 5 to the slow walk)
  double avg_degree;
  // column-blocked pass A (graphs whose Pc table outgrows the L2): col_blocks > 0 = number of column blocks, col_sub =
  // int[(cb_row_hi - cb_row_lo)][col_blocks + 1] sub-run boundaries (colblock_index_kernel) of the call's source rows
  int col_blocks = 0; const int* col_sub = nullptr; int64_t cb_row_lo = 0, cb_row_hi = 0;
  EdgeEncParams enc;
};

struct NodeProjParams {
  // fused first round: h = relu(bn(y_last)) computed on the fly (and stored to h0_out) instead of read from h_src
  const float* y_last; const double* y_stats; const float* y_gamma; const float* y_beta; double y_count; float* h0_out;
  int finalize_enc; EdgeEncParams enc; double e_total;   // block 0 also finalises the edge-encoder affines
  Drop drop; unsigned drop_stream;                       // dropout on the fused h0 (training)
  const float* h_src;        // [N][32]
  const float* h0;           // [N][32] (reattach_nodes) or nullptr
  const int* deg;            // mean aggregation: scale h_src rows by 1/max(deg,1); else nullptr
  const float* ue_w; int ue_ld;
  const float* un_w; int un_ld;
  int hn;                    // 32 or 64: width of [h0 | h]
  float* P; float* Q;
  float* zero_buf;           // [N][32] cleared for the coming aggregation, or nullptr
  int64_t n_nodes;           // N (global): the Pc half of P starts at P + 4 N
  int64_t node_begin, node_end;   // rows to project: [0, N), or the source rows of a row-complete edge shard
  // the node-only part of the node-update (z2) statistics, taken while Q is in registers: sum_i deg_i qb_ik and
  // sum_i deg_i qb_ik^2 with qb = Q + un_b and deg = the out-degree over the call's edges (pass B adds the edge part)
  const int* edge_deg; const float* un_b; double* z2_stats;   // z2_stats: this round's block at kRoundZ2Off, or nullptr
};

// Many-edge lists keep round 3's form of the node-update statistics: pass B adds per-node segment sums of e' into seg[N][4]
// (fp64 atomics per run and channel) and node_stat_kernel turns them, Q and the degrees into the sums -- an O(N) kernel of
// its own, but 15 vector instructions per 64 edges less in pass B, which is issue-bound at those sizes (same-box A/B at
// config 4: +0.7 % on the forward with the folded form, -4 % on the 150k-edge S02 graph).
struct NodeStatParams {
  const float* Q; const int* deg; double* seg;
  const float* un_w; const float* un_b; int un_ld; int un_eoff;
  double* stats;             // this round's block; z2 sums at kRoundZ2
  int64_t n_nodes;
  int64_t node_begin, node_end;   // as in NodeProjParams
};

struct GemmParams {
  const float* A; int64_t lda;       // [M][K] activations (raw pre-BN outputs of the previous layer, or x)
  const float* W;                    // [Nout][K]
  const float* bias;                 // [Nout]
  float* Y; int64_t ldy;             // [M][Nout] raw outputs (pre-BN)
  const double* stats_in;            // f64[2*K] column sum / sumsq of A's producer, or nullptr (layer 0)
  const float* gamma_in; const float* beta_in;
  double count;                      // BatchNorm row count (global N)
  double* stats_out;                 // f64[2*Nout], accumulated atomically
  int64_t M; int K; int Nout;
  Drop drop_in; unsigned drop_stream;  // training: dropout applied with the input BatchNorm+ReLU
  float* slab;                       // [split_k][M][Nout] scratch for split-K partial tiles, or nullptr
  int split_k;                       // set by launch_gemm_bn
  // fp16 two-piece path (gemm_bn_f16x3_kernel): |.|max bit patterns, u32[kAmaxRep] each (take the max), device
  // memory; nullptr = not available
  const unsigned* amax_a = nullptr;  // of A's source: x itself (no stats_in) or the producing layer's raw Y
  const unsigned* amax_w = nullptr;  // of W
  unsigned* amax_y = nullptr;        // out (atomicMax): of this layer's raw Y
  // Passenger workgroups (few-row graphs): the edge encoder's hidden-layer moments (enc2) ride in this GEMM's launch -- the
  // encoder chain and the edge branch are independent until the first round, and a few-row GEMM leaves most CUs idle.
  // launch_gemm_bn runs the job either way: inside the launch where the chosen kernel carries passengers, else behind it.
  int pass_blocks = 0; EdgeEncParams pass_enc = {}; const float* pass_attr = nullptr; int64_t pass_edges = 0;
  double pass_e_total = 0; double* pass_stat = nullptr;
};

// Few-row graphs (gemm_few.hip): layer 0 on pre-split operands, one workgroup per 64 x 32 tile over all of K
struct FewL0Params {
  const _Float16* Ah; const float* inv_a;   // planes [2][K/32][M][32] of x (k-tile-major, swizzled: lds_dma.h), [M] inverse row scales
  const _Float16* Wh; const float* inv_w;   // planes [2][K/32][Nout][32] of W0, [Nout]
  const float* bias;
  float* Y; int64_t ldy;
  double* stats_out;                        // f64[2*Nout], accumulated atomically (or nullptr)
  int64_t M; int K; int Nout;
};
// ... layers >= 1: 16 rows x 16 / 32 columns per workgroup, K cut between its waves
struct FewWaveParams {
  const float* A; int64_t lda;              // [M][K] raw Y of the previous layer
  const double* stats_in; const float* gamma_in; const float* beta_in; double count;   // its column statistics / BatchNorm
  const _Float16* Wh; const float* inv_w;   // planes [2][K/32][Nout][32], [Nout]
  const float* bias;
  float* Y; int64_t ldy;
  double* stats_out;                        // f64[2*Nout], accumulated atomically (or nullptr)
  int64_t M; int K; int Nout;
  Drop drop_in = {0, 0, 1.f, 0}; unsigned drop_stream = 0;   // training: Dropout applied with the input BatchNorm + ReLU
  int tiles = 0;                            // set by launch_few_wave
  // passenger workgroups as in GemmParams (the edge encoder's enc2 job rides in the last layer's launch)
  int pass_blocks = 0; EdgeEncParams pass_enc = {}; const float* pass_attr = nullptr; int64_t pass_edges = 0;
  double pass_e_total = 0; double* pass_stat = nullptr;
};
bool few_l0_shape(int K, int Nout);
bool few_wave_shape(int K, int Nout);
int few_wave_threads(int K);                // threads per workgroup launch_few_wave uses for a layer of depth K
bool few_rows_path(int64_t rows, int n_layers, const int* in_dim, const int* out_dim);
int launch_few_l0(const FewL0Params& p, hipStream_t s);       // 0 ok, 1 unsupported shape, MTMC_E_HIP
int launch_few_wave(const FewWaveParams& p, hipStream_t s);   // 0 ok, 1 unsupported shape

// First encoder layer on pre-split operands (gemm_presplit.hip): fp16 planes [2][rows][K] and one power-of-two
// inverse scale per row, written by launch_split_rows.
struct SplitGemmParams {
  const _Float16* Ah; const float* inv_a;   // [2][M][K], [M]
  const _Float16* Wh; const float* inv_w;   // [2][Nout][K], [Nout]
  const float* bias;
  float* Y; int64_t ldy;
  double* stats_out;                        // f64[2*Nout], accumulated atomically (or nullptr)
  unsigned* amax_y;                         // u32[kAmaxRep] (atomicMax) or nullptr
  int64_t M; int K; int Nout;
  // a launch over the row PANEL [m_lo, M) of planes that hold M_rows rows (pipelined layer 0: the operand split of the next
  // panel runs beside this panel's GEMM); defaults = the whole matrix.  bm: tile height to use (0 = chosen per launch)
  int64_t m_lo = 0; int64_t M_rows = 0; int bm = 0;
};
// Encoder layers >= 1 of many-row graphs (gemm_staged.hip): A = raw outputs of the previous layer (its BatchNorm + ReLU
// applied by producer waves on the way into LDS), W = the layer's weights pre-split by launch_split_rows.
struct StagedGemmParams {
  const float* A; int64_t lda;              // [M][K] raw Y of the previous layer
  const double* stats_in; const float* gamma_in; const float* beta_in; double count;   // its column statistics / BatchNorm
  const unsigned* amax_a;                   // u32[kAmaxRep]: its |Y|max
  const _Float16* Wh; const float* inv_w;   // [2][Nout][K] planes (k-tile-major, swizzled), [Nout] inverse row scales
  const float* bias;
  float* Y; int64_t ldy;
  double* stats_out;                        // f64[2*Nout], accumulated atomically
  unsigned* amax_y;                         // u32[kAmaxRep] (atomicMax) or nullptr
  int64_t M; int K; int Nout;
};
bool staged_layer(int64_t rows, int K, int Nout);       // many rows, K % 32 == 0, K <= 2048, Nout % 256 == 0 or (Nout == 128, rows >= 49152), not disabled
int launch_gemm_staged(const StagedGemmParams& p, hipStream_t s);   // 0 ok, 1 unsupported shape, MTMC_E_HIP
int staged_tile_rows(int64_t M, int tiles_n, int wm);   // (also used by the laboratory's second form, lab/staged2_lab.hip)
// The last, narrow encoder layers of many-row graphs as a row-streaming kernel (gemm_rows.hip): 128 -> 32
bool rows_layer(int64_t rows, int K, int Nout);
int launch_gemm_rows(const GemmParams& p, hipStream_t s);          // 0 ok, 1 unsupported shape
void launch_split_rows(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, hipStream_t s);
// rows [r_lo, r_hi) only, of planes laid out for `rows` rows (X points at row 0)
void launch_split_rows_range(const float* X, int64_t ld, int64_t rows, int K, void* H, float* inv, int64_t r_lo, int64_t r_hi,
                             hipStream_t s);
int presplit_tile_rows(int64_t M, int tiles_n);       // tile height the layer-0 GEMM picks for M rows (144..256)
// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device); false when HIP refuses (gemm_bn.hip)
bool allow_big_lds(const void* fn, int bytes);
bool presplit_layer0(int64_t rows, int K, int Nout);   // many rows, K % 64 == 0, K <= 2048, not disabled (MTMC_GEMM_NO_PRESPLIT)
int launch_gemm_presplit(const SplitGemmParams& p, hipStream_t s);   // 0 ok, 1 unsupported shape, MTMC_E_HIP

void launch_prep(const PrepParams& p, hipStream_t s);
void launch_enc2(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total, double* stat_enc2,
                 hipStream_t s);
void launch_pass_a(const RoundParams& p, hipStream_t s);
int plan_col_blocks(int64_t n_nodes, int64_t n_edges, double avg_degree, bool training);   // 0 = pass A in edge order
void launch_colblock_index(const RoundParams& p, int* sub, int B, int64_t row_lo, int64_t row_hi, hipStream_t s);
void launch_pass_b(const RoundParams& p, hipStream_t s);
void launch_pass_c(const RoundParams& p, hipStream_t s);
// Which pass-C kernel a call takes (host-only decision, also behind mtmc_mpn_plan): 0 = the half-wave walk; 1 = the
// matrix-core kernel with the walk launched behind it for unsorted rows (many-edge lists); 2 = the matrix-core kernel
// alone (few-edge lists).  avg_degree: edges per source row of THIS call's edges (RoundParams::avg_degree).
int plan_pass_c(int agg, bool deterministic, bool dropout, int64_t n_edges, int64_t n_nodes, double avg_degree);
bool pass_c_sorted_taken(int64_t n_nodes);
// Few-edge / many-edge regime of the edge passes (edge_kernels.hip: pick_ept).  Up to this many edges: one edge per thread in
// passes A / B, e' stored, the node-update statistics folded into node_proj + pass B, ONE any-order matrix-core pass C, the
// operand-split jobs and enc2 as passengers; above: four edges per thread, lazy e', node_stat, pass_c_sorted + the walk behind
// it.  Rounds 1-4: 2048 * 256 (one 2048-block grid of one edge per thread); round 5, after the few-edge forms lost three
// launches per round: 4-camera graphs of 0.75 / 1.08 / 1.47 / 1.92 / 3.0 M edges take 204 / 252 / 288 / 342 / 448 us with the
// few-edge forms and 223 / 252 / 296 / 315 / 375 us with the many-edge ones (profiles/r05_regime_sweep.txt): 6144 * 256.
#ifndef MTMC_SMALL_EDGES
#define MTMC_SMALL_EDGES (6144 * 256)
#endif
constexpr int64_t kSmallEdges = MTMC_SMALL_EDGES;
constexpr int64_t kDetSortedEdges = 2048 * 256;      // deterministic mode keeps pass_c_sorted_kernel<., true> from here up
void launch_seed_tick(unsigned long long* counter, unsigned long long* word, hipStream_t s);
int plan_edges_per_thread(int64_t n_edges);      // passes A / B: 1 on few-edge lists, 4 otherwise
void launch_classify_e0(const EdgeEncParams& enc, const float* attr, int64_t n_edges, double e_total,
                        const float* cls_w, const float* cls_b, int n_classes, float* logits, hipStream_t s);

void launch_node_proj(const NodeProjParams& p, hipStream_t s);
void launch_node_stat(const NodeStatParams& p, hipStream_t s);
// few-edge lists (one edge per thread in passes A / B): the node-update statistics come out of node_proj + pass B, no
// node_stat_kernel launch and no seg[] (RoundParams::fold_z2, NodeProjParams::z2_stats); many-edge lists: seg[] + node_stat
bool fold_node_stat(int64_t n_edges);
// h_dst[i][k] = relu(s_k * Y[i][k] + t_k) for local rows; stats over `count` rows
void launch_bn_relu_rows(const float* Y, int64_t ldy, int64_t rows, int dim, const double* stats, const float* gamma,
                         const float* beta, double count, float* dst, Drop drop, unsigned drop_stream, int64_t row0,
                         hipStream_t s, unsigned* amax_out = nullptr,    // amax_out: u32[kAmaxRep] |dst|max (atomicMax) or nullptr
                         float* dstT = nullptr, int64_t ldt = 0);         // dstT: dst^T [dim][ldt] as well, rows..ldt zero-filled
// dst = src (sum/max) or src / max(deg,1) (mean)
void launch_h_final(const float* src, const int* deg, int mean, int64_t n_nodes, float* dst, hipStream_t s);

int launch_gemm_bn(const GemmParams& p, hipStream_t s, int which = 3);   // returns 0 or MTMC_E_ARG
int gemm_plan(int64_t M, int K, int Nout, int* split_k, bool long_k_ok = false);

void launch_scatter(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                    float* out, float* count, int64_t* arg_out, int mode, hipStream_t s);

void launch_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t n_src, int64_t n_cols, int64_t dim_size,
                            int64_t* out, hipStream_t s);

size_t pp_workspace_bytes(int64_t n_nodes, int64_t n_edges, int64_t max_active);
int launch_postprocess(const float* logits, const int64_t* row, const int64_t* col, int64_t idx_stride, int64_t n_nodes,
                       int64_t n_edges, int num_cameras, int flags, int64_t max_active, float* prob1, int64_t* pred,
                       int64_t* id_pred, int32_t* info, void* workspace, size_t workspace_bytes, hipStream_t s);

}  // namespace mtmc
