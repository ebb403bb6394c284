/*
 * mtmc_mpn.h -- C ABI of the MI355X (gfx950) message-passing-network forward.
 *
 * This is the drop-in boundary for ONE hot path of
 * elun15/Graph-Convolutional-Network-for-Multi-Camera-Vehicle-Tracking: `MOTMPNet.forward`
 * (reference models/mpn.py:250-299).  Every pointer is a plain device pointer into HBM, every
 * size a plain integer; there are no torch (or any other framework) types in the signatures.
 * The host side that mirrors the reference's nn.Module interface (mtmc_mpn/modules.py) binds
 * these entry points with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Reference interface each entry point replaces:
 *   mtmc_mpn_forward           <- MOTMPNet.forward              models/mpn.py:250-299
 *     (node/edge encoders       MLPGraphIndependent.forward      models/mpn.py:128-142,
 *      MLP = Linear+BN+ReLU     MLP.forward                      models/mlp.py:11-33,
 *      one round                MetaLayer.forward                models/mpn.py:32-54,
 *      edge update              EdgeModel.forward                models/mpn.py:67-69,
 *      node update + aggregate  NodeModel.forward                models/mpn.py:97-99,
 *      classifier               MLPGraphIndependent.forward      models/mpn.py:291-292)
 *   mtmc_mpn_run_phase         <- the same, cut at the points where a multi-GPU run exchanges
 *                                 BatchNorm statistics / node states (no reference counterpart:
 *                                 the reference is single-GPU)
 *   mtmc_scatter_{add,mean,max}<- torch_scatter.scatter_{add,mean,max}(src, index, dim=0, dim_size)
 *                                 (third-party pytorch-scatter 2.0.8; call sites models/mpn.py:196,199,202)
 *   mtmc_mlp_forward           <- MLP.forward as a stand-alone op  models/mlp.py:32-33
 *   mtmc_build_graph           <- graph construction around the call  inference.py:402-456, train.py:316-342
 *   mtmc_postprocess           <- softmax/argmax + post_processing    inference.py:475-489, :70-169; utils.py:30-339
 *
 * All floating tensors are fp32, row-major, contiguous unless a stride argument says otherwise;
 * BatchNorm statistics are accumulated in fp64.  All work is enqueued on `stream` (a hipStream_t
 * passed as void*; NULL = the default stream); no entry point synchronises or allocates.
 * Return value: 0 on success, a negative MTMC_E_* code otherwise (mtmc_mpn_last_error() gives text).
 */
#ifndef MTMC_MPN_H
#define MTMC_MPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTMC_MPN_ABI_VERSION 6

#define MTMC_MAX_ENC_LAYERS 8   /* hidden layers of the node encoder MLP          */
#define MTMC_NODE_DIM 32        /* H : width of the node state the kernels are built for */
#define MTMC_EDGE_DIM 4         /* He: width of the edge state                    */
#define MTMC_MAX_CLASSES 4      /* classifier outputs per edge (reference: 2)     */

enum { MTMC_AGG_SUM = 0, MTMC_AGG_MEAN = 1, MTMC_AGG_MAX = 2 };

enum {
  MTMC_OK = 0,
  MTMC_E_ARG = -1,        /* bad dimension / NULL pointer / unsupported shape      */
  MTMC_E_WORKSPACE = -2,  /* workspace too small or misaligned                     */
  MTMC_E_HIP = -3,        /* a HIP runtime call failed                             */
  MTMC_E_ROWS = -4        /* fewer than 2 rows under a BatchNorm (the reference raises ValueError) */
};

/* One Linear(+BatchNorm affine) group: weight [out,in] row-major, bias [out], gamma/beta [out]. */
typedef struct mtmc_layer {
  const float* weight;
  const float* bias;
  const float* gamma;   /* BatchNorm1d.weight */
  const float* beta;    /* BatchNorm1d.bias   */
  int32_t in_dim;
  int32_t out_dim;
} mtmc_layer;

/* The 34 parameter tensors of the reference model + its structural flags. */
typedef struct mtmc_mpn_model {
  uint32_t struct_bytes;                        /* = sizeof(mtmc_mpn_model) of the header the CALLER was compiled against;
                                                   every entry point returns MTMC_E_ARG unless it equals the library's
                                                   (a binding written for an older, shorter struct fails loudly instead
                                                   of being read past its end) */
  int32_t n_enc_layers;                         /* node encoder: 2048->1024->512->128->32 => 4 */
  mtmc_layer enc_node[MTMC_MAX_ENC_LAYERS];     /* encoder.node_mlp.fc_layers.{0,4,8,12}(+1)   */
  mtmc_layer enc_edge[2];                       /* encoder.edge_mlp: in(1|2)->4->4              */
  mtmc_layer upd_edge;                          /* MPNet.edge_model.edge_mlp: [4, nf*64+ef*4]   */
  mtmc_layer upd_node;                          /* MPNet.node_model.node_mlp: [32, nf*32+4]     */
  mtmc_layer cls;                               /* classifier.edge_mlp.fc_layers.0: [C,4], no BN */
  int32_t agg;                                  /* MTMC_AGG_*  (node_agg_fn)                    */
  int32_t num_enc_steps;                        /* L                                            */
  int32_t num_class_steps;                      /* Cs                                           */
  int32_t reattach_nodes;                       /* reattach_initial_nodes                       */
  int32_t reattach_edges;                       /* reattach_initial_edges                       */
  float dropout_enc;                            /* Dropout p of both encoder MLPs (training only) */
  float dropout_upd_edge;                       /* ... of the edge-update MLP                     */
  float dropout_upd_node;                       /* ... of the node-update MLP                     */
} mtmc_mpn_model;

/* One forward call.  For a single GPU: node_lo=0, node_hi=n_nodes, n_edges_total=n_edges. */
typedef struct mtmc_mpn_call {
  uint32_t struct_bytes;     /* = sizeof(mtmc_mpn_call), checked like mtmc_mpn_model::struct_bytes */
  const float* x;            /* [node_hi-node_lo, F] rows node_lo.. of data.x           */
  int64_t x_row_stride;      /* elements between consecutive rows of x                  */
  const int64_t* row;        /* data.edge_index[0]: element e at row[e*idx_stride]      */
  const int64_t* col;        /* data.edge_index[1]                                      */
  int64_t idx_stride;        /* 1 for a contiguous [2,E]; 2 for the callers' [E,2].T view */
  const float* edge_attr;    /* [n_edges, edge_in_dim] contiguous                       */
  int64_t n_nodes;           /* N (global)                                              */
  int64_t n_edges;           /* E held by this call (the local shard)                   */
  int64_t n_edges_total;     /* E over all shards: the BatchNorm row count              */
  int64_t node_lo, node_hi;  /* node rows this call encodes                             */
  float* logits;             /* out [n_outputs][n_edges][C]; n_outputs = Cs, or 1 if L==0 */
  float* h_out;              /* out [n_nodes][32] = latent_node_feats                   */
  void* workspace;           /* >= mtmc_mpn_workspace_bytes(), 256-byte aligned         */
  size_t workspace_bytes;
  int32_t training;          /* 0: eval (Dropout = identity).  1: train: counter-based Dropout masks from `seed`,
                                every round keeps its own buffers so the workspace doubles as the backward tape
                                (size it with mtmc_mpn_train_workspace_bytes)                              */
  int32_t flags;             /* MTMC_F_*                                                */
  uint64_t seed;
  void* stream;              /* hipStream_t                                             */
  int64_t row_lo, row_hi;    /* multi-GPU, row-complete edge shards: MTMC_PH_ROUND_PROJ / _STAT work on node rows
                                [row_lo,row_hi) only -- the source rows of this call's edges; the host then exchanges
                                the column projections (mtmc_ws_layout.P_off) instead of the node state.  0,0 = all */
  void* weight_cache;        /* eval mode, optional (NULL = none): >= mtmc_mpn_weight_cache_bytes() of device memory, 256-byte
                                aligned, ZERO-FILLED once by the caller and then left alone between calls.  The library keeps
                                what it derives from the node-encoder WEIGHTS alone in it (fp16 operand planes, row scales) and
                                verifies it against the weights' CONTENT on every call -- a 64-bit fingerprint per 8 weight rows,
                                taken on the device -- so the caller promises nothing about the weights: a changed weight is
                                split again, an unchanged one is not.  One cache per stream that runs forwards concurrently.
                                Few-row graphs need it for the kernels of csrc/gemm_few.hip (without: the split-K kernels).   */
  size_t weight_cache_bytes;
} mtmc_mpn_call;

#define MTMC_F_DETERMINISTIC 1   /* row-sorted edge lists: sum/mean aggregation through per-chunk partials added in
                                    a fixed order instead of float atomics (bitwise run-to-run reproducible h) */
#define MTMC_F_GLOBAL_DEG 4      /* mean aggregation divides by workspace deg_global (multi-GPU) instead of deg */
/* (8 was MTMC_F_WEIGHTS_CACHED in ABI v5: a host-side promise that the weights had not changed.  Gone: the weight-plane
 * cache verifies itself, mtmc_mpn_call::weight_cache.) */
#define MTMC_F_FORK 2            /* run the edge branch (prep, edge-encoder moments) on an internal side stream
                                    beside the node-encoder GEMMs, joined by events (default: one stream) */
#define MTMC_F_SEED_ON_DEVICE 16 /* training: `seed` holds the ADDRESS of a uint64 counter in device memory (8-byte aligned).  The
                                    forward takes the counter's value as this call's Dropout seed, stores it in the workspace
                                    (= the tape: the backward of the same tape reads it there) and moves the counter on by one --
                                    on the stream, so a HIP graph that holds a whole training step (forward, loss, backward,
                                    optimizer) draws new masks on every replay.  Additive in ABI v6: calls without the bit are
                                    as before (`seed` by value).  Reference: nn.Dropout's generator state, models/mlp.py:20-21 */

/* Byte offsets inside the workspace of the regions a multi-GPU host exchanges between phases.
 * Every statistics block is kept in MTMC_STAT_REPLICAS replicas (consumers add them up), so a host simply
 * all-reduces the whole replicated block.  Strides are in doubles. */
#define MTMC_STAT_REPLICAS 16
#define MTMC_ATTR_STRIDE 16      /* edge_attr moments : m1[2] | m2 packed upper triangle[3]                */
#define MTMC_ENC2_STRIDE 16      /* hidden edge-encoder moments : m1[4] | m2 packed[10]                     */
#define MTMC_Z1_STRIDE 16        /* edge-update pre-activation : sum[4] | sumsq[4]                          */
#define MTMC_M_STRIDE 16         /* updated edge state e' : m1[4] | m2 packed[10]                           */
#define MTMC_Z2_STRIDE 64        /* node-update pre-activation : sum[32] | sumsq[32] (without the A M A^T term) */
#define MTMC_ROUND_BLOCK (MTMC_STAT_REPLICAS * (MTMC_Z1_STRIDE + MTMC_M_STRIDE + MTMC_Z2_STRIDE))
typedef struct mtmc_ws_layout {
  size_t total_bytes;
  size_t zero_bytes;         /* the leading region [0, zero_bytes) is cleared by MTMC_PH_BEGIN           */
  size_t flags_off;          /* int32[8]: [0] rows out of order, [1] indices out of range                */
  size_t stat_attr_off;      /* f64[REPLICAS][ATTR_STRIDE]; DIRECTLY followed by encoder layer 0's block: a multi-GPU
                                host all-reduces [stat_attr_off, stat_enc_layer_off[0] + 16 * out_dim_0) as ONE message
                                after MTMC_PH_NODE_COMBINE 0 (the edge branch's statistics ride with the node encoder's) */
  size_t stat_enc2_off;      /* f64[REPLICAS][ENC2_STRIDE]; directly followed by encoder layer 1's block (if there is one) */
  size_t stat_enc_layer_off[MTMC_MAX_ENC_LAYERS];  /* per encoder layer l: f64[2][out_dim_l] column sum | sumsq    */
  size_t stat_round_off;     /* f64[L][ROUND_BLOCK]: per round the z1, e'-moment and z2 blocks in order   */
  size_t deg_off;            /* i32[N]  out-degree of the LOCAL edges (row histogram)                     */
  size_t seg_off;            /* f64[N][4] per-node segment sums of e' over the LOCAL edges (many-edge lists; zero bytes
                                on few-edge lists, whose pass B adds the statistics' edge part itself)            */
  size_t h0_off;             /* f32[N][32] encoded node state (rows node_lo..node_hi written locally)      */
  size_t h_acc_off[2];       /* f32[N][32] x2 aggregation ping-pong: round r aggregates into [r & 1],
                                except that the last round of a sum/max model aggregates into h_out       */
  size_t deg_global_off;     /* i32[N]  degree over all shards; read instead of deg for mean aggregation
                                when MTMC_F_GLOBAL_DEG is set (the host fills it)                         */
  size_t P_off;              /* f32[2][N][4] the round's edge-update projections: Pr rows, then Pc rows (gathered
                                by column: every shard needs ALL of Pc, 16 B per node, and only its own Pr / Q)   */
} mtmc_ws_layout;

enum {                       /* phases in forward order; `arg` = encoder layer or round (0-based) */
  MTMC_PH_BEGIN = 0,         /* clear statistics; int64 -> int32 indices; degree; attr moments   */
  MTMC_PH_EDGE_ENC = 1,      /* moments of the edge-encoder hidden layer                         */
  MTMC_PH_NODE_ENC = 2,      /* arg = layer: GEMM (+ column statistics unless split along K)      */
  MTMC_PH_NODE_H0 = 3,       /* h0 = relu(bn(Y_last)) for rows [node_lo,node_hi)                 */
  MTMC_PH_ROUND_PROJ = 4,    /* arg = round: per-node projections Pr|Pc|Q, clear next h buffer   */
  MTMC_PH_ROUND_A = 5,       /* statistics of the edge-update pre-activation                     */
  MTMC_PH_ROUND_B = 6,       /* e' (stored), its moments and per-node segment sums               */
  MTMC_PH_ROUND_STAT = 7,    /* statistics of the node-update pre-activation by moments (many-edge lists; a no-op on
                                few-edge lists: complete after MTMC_PH_ROUND_PROJ + MTMC_PH_ROUND_B there)      */
  MTMC_PH_ROUND_C = 8,       /* messages, aggregation into h, classifier logits                  */
  MTMC_PH_END = 9,           /* mean scaling / copy of the final node state to h_out             */
  MTMC_PH_NODE_COMBINE = 10  /* arg = layer, right after MTMC_PH_NODE_ENC: sums the split-K slabs of a few-row
                                layer and takes its column statistics (no-op when the layer was not split) */
};

int32_t mtmc_mpn_abi_version(void);
const char* mtmc_mpn_last_error(void);
size_t mtmc_mpn_workspace_bytes(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges);
size_t mtmc_mpn_weight_cache_bytes(const mtmc_mpn_model* model);   /* size of mtmc_mpn_call::weight_cache (0: bad model) */
int32_t mtmc_mpn_workspace_layout(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges,
                                  mtmc_ws_layout* out);
int32_t mtmc_mpn_forward(const mtmc_mpn_model* model, const mtmc_mpn_call* call);

/* Training (reference train.py:356,424: outputs, _ = model(data); loss.backward()).
 * Forward: mtmc_mpn_forward with call->training = 1 and a workspace of mtmc_mpn_train_workspace_bytes().
 * Backward: the SAME model/call (same workspace, untouched since the forward) plus the incoming gradients
 *   d_logits [n_outputs][E][C] and d_h [N][32] (either may be NULL = zero).  `grads` has the layout of the model
 *   struct; its weight/bias/gamma/beta pointers name the buffers that RECEIVE the 34 parameter gradients
 *   (overwritten, fp32).  d_x [N][F] / d_edge_attr [E][Fe] are written if non-NULL. */
size_t mtmc_mpn_train_workspace_bytes(const mtmc_mpn_model* model, int64_t n_nodes, int64_t n_edges);
int32_t mtmc_mpn_backward(const mtmc_mpn_model* model, const mtmc_mpn_call* call, const float* d_logits,
                          const float* d_h, const mtmc_mpn_model* grads, float* d_x, float* d_edge_attr);
/* The same with one gradient pointer per classified step (host array of min(Cs, L) device pointers, NULL entries =
 * no gradient for that step; what torch.autograd hands over) and, optionally, the ONE buffer all gradient tensors
 * of `grads` were carved from: it is cleared with a single memset instead of one per tensor. */
int32_t mtmc_mpn_backward_steps(const mtmc_mpn_model* model, const mtmc_mpn_call* call,
                                const float* const* d_logits_steps, const float* d_h, const mtmc_mpn_model* grads,
                                void* grads_flat, size_t grads_flat_bytes, float* d_x, float* d_edge_attr);
/* The same with the gradient carving done by the library: `flat` receives every parameter gradient in struct order
 * (node encoder layers, edge encoder, edge update, node update, classifier; weight, bias, then gamma, beta where the layer
 * has a BatchNorm), each piece on a 64-float boundary; mtmc_mpn_grad_layout writes the piece offsets (in floats, up to
 * max_offsets of them) and returns the total length.  One allocation and no second pointer struct per call. */
int64_t mtmc_mpn_grad_layout(const mtmc_mpn_model* model, int64_t* offsets, int32_t max_offsets);
int32_t mtmc_mpn_backward_flat(const mtmc_mpn_model* model, const mtmc_mpn_call* call,
                               const float* const* d_logits_steps, const float* d_h, float* flat, int64_t flat_floats,
                               float* d_x, float* d_edge_attr);
int32_t mtmc_mpn_run_phase(const mtmc_mpn_model* model, const mtmc_mpn_call* call, int32_t phase, int32_t arg);
/* n consecutive (phase, arg) pairs in ONE call (phase_args[2*i], phase_args[2*i+1]): what a multi-GPU host runs between two
 * of its collectives, without one library call (and one argument check) per phase. */
int32_t mtmc_mpn_run_phases(const mtmc_mpn_model* model, const mtmc_mpn_call* call, const int32_t* phase_args, int32_t n);

/* Which kernels a call would run -- a host-only query (nothing is launched, no pointer of `call` is read: only its
 * sizes, ranges, flags and `training`), so that a multi-GPU host or a test can check that a SHARD takes the kernels the
 * whole graph would.  Encoder layers are planned for the node_hi - node_lo rows the call encodes. */
enum { MTMC_GEMM_GENERIC = 0, MTMC_GEMM_INLOOP_64 = 1, MTMC_GEMM_INLOOP_128 = 2, MTMC_GEMM_PRESPLIT_256 = 3,
       MTMC_GEMM_STAGED_128 = 4, MTMC_GEMM_ROWS_16 = 5, MTMC_GEMM_FEW_L0 = 6, MTMC_GEMM_FEW_WAVE = 7 };
enum { MTMC_PASS_C_WALK = 0, MTMC_PASS_C_MFMA_SORTED = 1, MTMC_PASS_C_MFMA_ANY = 2 };
typedef struct mtmc_mpn_plan {
  int32_t enc_kernel[MTMC_MAX_ENC_LAYERS];   /* MTMC_GEMM_*: one-thread-per-output fallback / in-loop operand split on
                                                64x64 or 128x128 tiles / pre-split fp16 planes + 256x256 tiles / (layers
                                                >= 1 of many-row graphs) 128 x 256 or 256 x 128 tiles staged by producer waves /
                                                (narrow last layers of many-row graphs) one wave per 16 rows /
                                                (few-row graphs with a weight-plane cache: the call's weight_cache pointer is
                                                tested for NULL, never read) layer 0 on pre-split operands in 64 x 32 tiles over
                                                all of K / later layers with K cut between the waves of a workgroup   */
  int32_t enc_split_k[MTMC_MAX_ENC_LAYERS];  /* K slices (few-row layers; > 1 => MTMC_PH_NODE_COMBINE does work)    */
  int32_t edges_per_thread;                  /* passes A / B                                                        */
  int32_t lazy_edges;                        /* 1: e' is never stored, consumers recompute it from z1               */
  int32_t pass_c;                            /* MTMC_PASS_C_*: half-wave walk / pass_c_sorted_kernel (many-edge lists; it
                                                returns at once on unsorted rows and the walk behind it does the round) /
                                                the any-order matrix-core kernel (few-edge lists: alone; many-edge lists
                                                of >= 2^24 global node rows: returns on unsorted rows like the sorted one) */
  double avg_degree;                         /* edges per source row the pass-C choice was made on                  */
  int32_t pass_a_col_blocks;                 /* 0: pass A walks the edges in order; B > 0: by B column blocks, so that the
                                                gathered projections Pc of a block stay in an XCD's L2 (graphs whose 16 B / node
                                                table outgrows it; row-sorted lists with ascending columns -- decided on the
                                                device, the in-order kernel is launched behind it)                     */
  int32_t layer0_panels;                     /* mtmc_mpn_forward only: row panels of the pre-split layer 0 (the operand split of
                                                panel i + 1 runs on a side stream beside panel i's GEMM); 1 = one split pass, one
                                                GEMM (also what mtmc_mpn_run_phase(s) does)                              */
  int32_t enc2_passenger;                    /* mtmc_mpn_forward only: 1 = MTMC_PH_EDGE_ENC's work rides as passenger workgroups
                                                in the last node-encoder layer's launch (few-row, few-edge graphs)       */
  int32_t node_stat_folded;                  /* 1 = few-edge list: the node-update statistics come out of MTMC_PH_ROUND_PROJ +
                                                MTMC_PH_ROUND_B, MTMC_PH_ROUND_STAT launches nothing                     */
} mtmc_mpn_plan;
int32_t mtmc_mpn_plan_call(const mtmc_mpn_model* model, const mtmc_mpn_call* call, mtmc_mpn_plan* out);

/* out[dim_size, C] (fp32) <- scatter of src[E, C] by index[E] along dim 0; rows nobody writes are 0.
 * mean divides by max(count,1); max also writes arg_out[dim_size, C] (int64, E where untouched) if non-NULL. */
int32_t mtmc_scatter_add(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                         int64_t dim_size, float* out, void* stream);
/* the integer call form of the reference's post-processing, scatter_add(int64 [E], index, dim_size=N)
 * (reference utils.py:173-174, :308-309): int64 in, int64 out, exact */
int32_t mtmc_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                             int64_t dim_size, int64_t* out, void* stream);
int32_t mtmc_scatter_mean(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                          int64_t dim_size, float* out, float* count_scratch, void* stream);
int32_t mtmc_scatter_max(const float* src, const int64_t* index, int64_t n_src, int64_t n_cols,
                         int64_t dim_size, float* out, int64_t* arg_out, void* stream);

/* y[rows, out] = relu(batchnorm_batchstats(x[rows,in] . W^T + b)) for one Linear+BN+ReLU group
 * (gamma == NULL: bare Linear).  stats_scratch: f64[2*out], y_raw aliases y. */
int32_t mtmc_mlp_layer_forward(const mtmc_layer* layer, const float* x, int64_t x_row_stride, int64_t rows,
                               float* y, double* stats_scratch, void* stream);

/* Diagnostics / unit tests: Y[M][N] = A[M][K] . W[N][K]^T + bias through the node encoder's GEMM dispatch exactly as the
 * forward runs its first layer (operand scales gathered on the device, fp16 two-piece kernel where it applies);
 * K a multiple of 32.  scratch: u32[48]; stats: f64[2*N] column sum / sum of squares of Y, or NULL. */
int32_t mtmc_linear_raw(const float* A, int64_t lda, const float* W, const float* bias, float* Y, int64_t M, int32_t K,
                        int32_t N, uint32_t* scratch, double* stats, void* stream);

/* One encoder layer as the forward runs it on FEW-ROW graphs (csrc/gemm_few.hip): stats_in == NULL: layer 0,
 * Y = A . W^T + bias with both operands split into fp16 planes first (K a multiple of 64, <= 2048; N a multiple of 32);
 * else a later layer, Y = relu(bn(A)) . W^T + bias as in mtmc_linear_staged_raw (K a multiple of 32; N a multiple of 16).
 * work: >= 4*M*K + 4*N*K + 4*(M+N) + 1024 bytes; stats: f64[2*N] or NULL. */
int32_t mtmc_linear_few_raw(const float* A, int64_t lda, const double* stats_in, const float* gamma_in, const float* beta_in,
                            double count, const float* W, const float* bias, float* Y, int64_t M, int32_t K, int32_t N,
                            void* work, uint64_t work_bytes, double* stats, void* stream);

/* One encoder layer >= 1 as the forward runs it on many-row graphs (csrc/gemm_staged.hip: role-split kernel, N a multiple
 * of 256 or N = 128; csrc/gemm_rows.hip: row-streaming kernel for the narrow last layer, K = 128 and N = 32):
 * Y[M][N] = relu(bn(A))[M][K] . W[N][K]^T + bias, bn = BatchNorm1d with batch statistics given as stats_in = f64 column
 * sum[K] | sum of squares[K] over `count` rows, and gamma_in / beta_in [K] (reference models/mlp.py:14-27: the previous
 * group's BatchNorm + ReLU fused into this Linear).  K a multiple of 32 in [64, 2048].
 * work: >= 4*N*K + 4*N + 256 bytes; scratch: u32[48]; stats: f64[2*N] column sum / sum of squares of Y, or NULL. */
int32_t mtmc_linear_staged_raw(const float* A, int64_t lda, const double* stats_in, const float* gamma_in,
                               const float* beta_in, double count, const float* W, const float* bias, float* Y, int64_t M,
                               int32_t K, int32_t N, void* work, uint64_t work_bytes, uint32_t* scratch, double* stats,
                               void* stream);

/* The pre-split twin of mtmc_linear_raw (csrc/gemm_presplit.hip; what the forward runs for encoder layer 0 of graphs
 * with >= 4096 nodes): A and W are first split into fp16 pairs with one power-of-two scale per row, stored k-tile-major
 * (work: >= 4*M*K + 4*N*K + 4*(M+N) + 1024 bytes of device memory), then multiplied by a plain fp16 MFMA GEMM fed by
 * LDS-DMA.  K a multiple of 64, K <= 2048.  reuse_planes != 0: the planes a previous call left in `work` are multiplied
 * again (times the GEMM alone).  (The kernels this one was chosen against are not in this library: csrc/lab/.) */
int32_t mtmc_linear_presplit_raw(const float* A, int64_t lda, const float* W, const float* bias, float* Y, int64_t M,
                                 int32_t K, int32_t N, void* work, uint64_t work_bytes, uint32_t* scratch,
                                 double* stats, int32_t reuse_planes, void* stream);

/* ---- graph construction (SURVEY.md 8(f)-1,2; replaces reference inference.py:402-456 / train.py:316-342) ----
 * feats [N][F] raw per-tracklet features -> x_out [N][F] (column-normalised like F.normalize(dim=0) if l2norm),
 * edge_index_out [E][2] int64 (row, col) in the reference's order (per camera, ascending: nodes of the camera x nodes
 * outside it; its transpose view is the [2,E] tensor the callers pass on), edge_attr_out [E][2] =
 * [pairwise_distance, 1 - cosine_similarity], edge_labels_out [E] (1.0 iff node_labels match; NULL to skip).
 * The camera structure comes as small device arrays the host derives from the camera ids:
 *   in_list/in_off[n_cams+1]: nodes of camera c;  out_list/out_off[n_cams+1]: nodes not in c (ascending);
 *   block_off[n_cams+1]: first edge of camera c's block, block_off[n_cams] = E.
 * Uses one N x N Gram matrix on the matrix cores instead of per-edge 2048-d gathers; N <= 46000. */
size_t mtmc_graph_workspace_bytes(int64_t n_nodes, int32_t feat_dim);
int32_t mtmc_build_graph(const float* feats, int64_t feat_row_stride, int64_t n_nodes, int32_t feat_dim, int32_t l2norm,
                         const int32_t* in_list, const int32_t* in_off, const int32_t* out_list, const int64_t* out_off,
                         const int64_t* block_off, int32_t n_cams, int64_t n_edges, const int64_t* node_labels,
                         float* x_out, int64_t* edge_index_out, float* edge_attr_out, float* edge_labels_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- the training callers' loss: F.cross_entropy / nn.CrossEntropyLoss(weight, reduction) on [n][n_classes<=4]
 * logits (reference train.py:88-93, :109-142, :178-186).  mode 0 = mean, 1 = sum, 2 = none.
 * forward: per_sample[i] = w[y_i] * (logsumexp(x_i) - x_i[y_i]) (NULL to skip); sums = f64[2*MTMC_STAT_REPLICAS]
 * scratch, on return sums[0] = sum_i per_sample[i], sums[1] = sum_i w[y_i]; loss_out[0] = sums[0]/sums[1] (mean) or
 * sums[0] (NULL to skip).  Rows with y_i == ignore_index count for nothing.
 * backward: d_logits[i][c] = g_i * w[y_i] * (softmax(x_i)[c] - [c == y_i]) with g_i = grad[0]/sums[1] (mean, needs the
 * forward's sums), grad[0] (sum) or grad[i] (none).  All pointers are device pointers. */
int32_t mtmc_cross_entropy_forward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                   int32_t n_classes, int64_t ignore_index, int32_t mode, float* per_sample,
                                   double* sums, float* loss_out, void* stream);
int32_t mtmc_cross_entropy_backward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                    int32_t n_classes, int64_t ignore_index, int32_t mode, const float* grad,
                                    const double* sums, float* d_logits, void* stream);

/* sum over the classified steps of cross_entropy(step, labels) in one pass (the training loop's loss, reference
 * train.py:118-138): logits = the contiguous [n_steps][n][n_classes] block of one forward, labels [n] shared by the steps.
 * mode 0 = mean per step, 1 = sum; sums as above (of all steps together); loss_out[0] = the summed loss.
 * backward: d_logits [n_steps][n][n_classes] for grad[0] = d loss. */
int32_t mtmc_cross_entropy_steps_forward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                         int32_t n_classes, int32_t n_steps, int64_t ignore_index, int32_t mode,
                                         double* sums, float* loss_out, void* stream);
int32_t mtmc_cross_entropy_steps_backward(const float* logits, const int64_t* labels, const float* weight, int64_t n,
                                          int32_t n_classes, int32_t n_steps, int64_t ignore_index, int32_t mode,
                                          const float* grad, const double* sums, float* d_logits, void* stream);

/* counts[4] (int64, device) = {TP, FP, TN, FN} of argmax(logits [n][n_classes]) against 0/1 labels [n], in one pass:
 * the confusion counts the training / validation loops compute with boolean-mask indexing (reference train.py:98-107,
 * inference.py:20-67).  Labels other than 0 / 1 are skipped; prediction = (argmax == 1), first maximum on ties. */
int32_t mtmc_edge_confusion(const float* logits, const int64_t* labels, int64_t n, int32_t n_classes, int64_t* counts,
                            void* stream);

/* ---- post-processing of the last logits (replaces reference inference.py:475-489, post_processing
 * inference.py:70-169 and utils.py compute_SCC_and_Clusters :30-52, splitting :54-123, remove_edges_single_direction
 * :125-142, pruning :144-339) ----
 * logits [E][2] -> prob1 [E] = softmax(logits)[:,1] and predictions [E] = argmax (both written); then, per `flags`,
 * symmetric cut, flow pruning (at most num_cameras-1 active out-/in-edges per node), second cut, splitting of clusters
 * with more than num_cameras nodes; predictions is updated in place and id_pred [N] receives the cluster number of
 * every node in the reference's numbering (networkx SCC emission order, stably sorted by size, isolated nodes last).
 * logits == NULL: prob1 and predictions are INPUTS (reproduces a run from given probabilities exactly).
 * info [8] (int32, device): active edges in, active edges out, clusters, status (0 ok, 1 = more than max_active
 * active edges: predictions left at argmax, 2 = over-sized cluster without an active edge, 3 = splitting iteration cap
 * reached, 4 = an active edge named a node outside [0, n_nodes): clamped), splitting iterations,
 * component walks, pruning rounds, 100 MHz ticks spent in the walks.  max_active <= 0: size the workspace for E active edges.
 * row/col: int64 with element stride idx_stride, as in struct mtmc_mpn_call.  Nothing is synchronised or allocated. */
enum { MTMC_PP_CUTTING = 1, MTMC_PP_PRUNING = 2, MTMC_PP_SPLITTING = 4 };
size_t mtmc_postprocess_workspace_bytes(int64_t n_nodes, int64_t n_edges, int64_t max_active);
int32_t mtmc_postprocess(const float* logits, const int64_t* row, const int64_t* col, int64_t idx_stride,
                         int64_t n_nodes, int64_t n_edges, int32_t num_cameras, int32_t flags, int64_t max_active,
                         float* prob1, int64_t* predictions, int64_t* id_pred, int32_t* info,
                         void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTMC_MPN_H */
