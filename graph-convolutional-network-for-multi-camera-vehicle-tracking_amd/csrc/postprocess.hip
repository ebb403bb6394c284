// Post-processing of the MPN's edge logits on the device.
//
// Replaces reference inference.py:475-489 + post_processing (:70-169) + utils.py compute_SCC_and_Clusters (:30-52),
// splitting (:54-123), remove_edges_single_direction (:125-142), pruning (:144-339): softmax / argmax over the E
// edges, then -- on the few edges predicted active -- symmetric cut, flow pruning, strongly connected components and
// splitting of over-sized clusters, producing the same `predictions [E]` and the same cluster numbering `ID_pred [N]`.
//
// Shape of the work: E-sized, embarrassingly parallel and HBM-bound up to the compaction of the A active edges
// (A ~ 1 % of E on a trained model); everything after that is a data-dependent loop nest on A edges / N nodes whose
// iteration counts are only known on the device.  So:
//   pp_classify_kernel   wide: p1 = softmax(logits)[:,1], prediction = argmax (ties -> 0), per-block active counts
//   pp_scan_kernel       one block: exclusive scan of the block counts -> A
//   pp_compact_kernel    wide: active edges, in ascending edge order, to (edge id, u, v, p1) arrays
//   pp_graph_kernel      ONE workgroup of 1024 lanes, resident for the whole cut / prune / cut / split sequence:
//                        set-parallel steps run on all lanes with block barriers in between; the cluster *numbering*
//                        is defined by networkx's depth-first traversal order (utils.py:31), which is inherently
//                        sequential, so that walk runs on lane 0 -- out of LDS whenever N <= 2048 and A <= 16384.
// No host round trip, no dynamic allocation: a caller can enqueue it right behind the forward.
#include "kernels.h"
#include "../../include/mtmc_mpn.h"

#include <stdint.h>

namespace mtmc {

constexpr int kPpChunk = 1024;            // edges per block in the wide kernels (256 lanes x 4)
constexpr int kPpThreads = 1024;          // the graph kernel's workgroup
constexpr int kPpLdsNodes = 2048;         // LDS-resident graph state up to this many nodes ...
constexpr int kPpLdsEdges = 8192;         // ... and this many active edges
constexpr int kPpNodeArrays = 13;         // rowptr, label, t0..t6, component records (size, tree key, sequence, free ids)
constexpr unsigned kDead = 0x80000000u;
constexpr int kPpMaxSplitIters = 1 << 18;  // splitting iterations before giving up (status 3); real runs need tens

struct PpParams {
  const float* logits;                    // [E][2] or nullptr (then prob1/pred are inputs)
  const int64_t* row; const int64_t* col; int64_t idx_stride;
  int64_t n_nodes, n_edges;
  int num_cameras, flags;
  float* prob1;                           // [E]
  int64_t* pred;                          // [E]
  int64_t* id_pred;                       // [N]
  int32_t* info;                          // [8]: A_in, A_out, clusters, status, split iterations, scc walks, prune rounds
  // workspace
  int* hdr; int* block_count; int* a_idx; int* a_u; int* a_v; float* a_p; int* a_slot; unsigned char* alive;
  unsigned char* mark; int* g_node; unsigned* g_csr; int* g_flags; int* b_idx; int* b_u; int* b_v; float* b_p; int64_t cap;
  int* g_wcc; int* g_dirty;                // [N] each: weakly connected component of a node, per-component dirty flag
};

__device__ __forceinline__ float softmax_p1(float l0, float l1) {
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  return e1 / (e0 + e1);
}

__global__ __launch_bounds__(256) void pp_classify_kernel(PpParams p) {
  __shared__ int wsum[4];
  const int64_t base = (int64_t)blockIdx.x * kPpChunk;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + i * 256 + threadIdx.x;
    int on = 0;
    if (e < p.n_edges) {
      if (p.logits) {
        const float2 l = reinterpret_cast<const float2*>(p.logits)[e];
        p.prob1[e] = softmax_p1(l.x, l.y);
        on = l.y > l.x ? 1 : 0;                         // torch.argmax: first maximum
        p.pred[e] = on;
      } else {
        on = p.pred[e] == 1;
      }
    }
    cnt += __popcll(__ballot(on));
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) p.block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of n ints in place, total returned to every lane (one block, any n)
__device__ int block_exclusive_scan(int* a, int64_t n, int* sh /* [blockDim.x/64 + 1] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int carry = 0;
  for (int64_t base = 0; base < n; base += blockDim.x) {
    const int64_t i = base + threadIdx.x;
    const int v = i < n ? a[i] : 0;
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(s, off, 64);
      if (lane >= off) s += t;
    }
    if (lane == 63) sh[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0;
      for (int w = 0; w < nw; ++w) { const int t = sh[w]; sh[w] = acc; acc += t; }
      sh[nw] = acc;
    }
    __syncthreads();
    if (i < n) a[i] = carry + sh[wid] + s - v;
    carry += sh[nw];
    __syncthreads();
  }
  return carry;
}

__global__ __launch_bounds__(1024) void pp_scan_kernel(PpParams p, int n_blocks) {
  __shared__ int sh[17];
  const int total = block_exclusive_scan(p.block_count, n_blocks, sh);
  if (threadIdx.x == 0) {
    p.hdr[0] = total;
    p.hdr[2] = 0;
    p.hdr[3] = 0;
    p.hdr[1] = total > p.cap ? 1 : 0;                   // more active edges than the workspace was sized for
    p.info[0] = total;
  }
}

__global__ __launch_bounds__(256) void pp_compact_kernel(PpParams p) {
  __shared__ int woff[4];
  if (p.hdr[1]) return;
  const int64_t base = (int64_t)blockIdx.x * kPpChunk;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int out = p.block_count[blockIdx.x];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + i * 256 + threadIdx.x;
    const int on = (e < p.n_edges && p.pred[e] == 1) ? 1 : 0;
    const unsigned long long b = __ballot(on);
    if (lane == 0) woff[wid] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wid; ++w) before += woff[w];
    const int tot = woff[0] + woff[1] + woff[2] + woff[3];
    if (on) {
      const int k = out + before + __popcll(b & ((1ull << lane) - 1));
      p.a_idx[k] = (int)e;
      int64_t u = p.row[e * p.idx_stride], v = p.col[e * p.idx_stride];
      if (u < 0 || u >= p.n_nodes || v < 0 || v >= p.n_nodes) {   // never index with it: clamp and report (status 4)
        p.hdr[3] = 1;
        u = u < 0 ? 0 : (u >= p.n_nodes ? p.n_nodes - 1 : u);
        v = v < 0 ? 0 : (v >= p.n_nodes ? p.n_nodes - 1 : v);
      }
      p.a_u[k] = (int)u;
      p.a_v[k] = (int)v;
      p.a_p[k] = p.prob1[e];
      p.alive[k] = 1;
    }
    out += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// the resident workgroup
// ------------------------------------------------------------------------------------------------
struct PpGraph {
  // N-sized
  int* rowptr;     // [N+1] CSR over ALL compacted active edges, rows in ascending active id (= edge order)
  int* label;      // [N]   cluster number of the node's component (output numbering)
  int* t0; int* t1; int* t2; int* t3; int* t4; int* t5; int* t6;   // [N] each, phase-dependent (see uses)
  int* csize; int* ctree; int* cseq; int* freel;                  // [N] each: component records of the splitting loop
  unsigned* csr;   // [A]   target node | kDead
  int n, a, cams;
};

__device__ __forceinline__ void pp_kill(const PpParams& p, const PpGraph& g, int i) {
  p.alive[i] = 0;
  g.csr[p.a_slot[i]] |= kDead;
}

// CSR of the active edges by source node; each row in ascending active id, i.e. in the order networkx inserts the
// successors when the reference builds nx.DiGraph(list of active edges).
__device__ void pp_build_csr(const PpParams& p, const PpGraph& g, int* sh) {
  int* deg = g.rowptr;
  for (int i = threadIdx.x; i <= g.n; i += blockDim.x) deg[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) atomicAdd(&deg[p.a_u[i]], 1);
  __syncthreads();
  block_exclusive_scan(deg, (int64_t)g.n + 1, sh);      // rowptr[n] = A
  int* fill = g.t0;
  for (int i = threadIdx.x; i < g.n; i += blockDim.x) fill[i] = g.rowptr[i];
  __syncthreads();
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) p.a_slot[i] = atomicAdd(&fill[p.a_u[i]], 1);   // any order ...
  __syncthreads();
  // ... then each row's slots re-dealt in ascending active id (rows are short: insertion sort per row)
  int* ids = reinterpret_cast<int*>(g.csr);
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) ids[p.a_slot[i]] = i;
  __syncthreads();
  for (int r = threadIdx.x; r < g.n; r += blockDim.x) {
    const int lo = g.rowptr[r], hi = g.rowptr[r + 1];
    for (int x = lo + 1; x < hi; ++x) {
      const int key = ids[x];
      int y = x - 1;
      while (y >= lo && ids[y] > key) { ids[y + 1] = ids[y]; --y; }
      ids[y + 1] = key;
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < g.a; s += blockDim.x) {
    const int i = ids[s];
    p.a_slot[i] = s;
    p.mark[s] = 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) g.csr[p.a_slot[i]] = (unsigned)p.a_v[i];
  __syncthreads();
}

// utils.py:125-142: an active edge survives only if its reverse is active too -- decided on a snapshot
__device__ void pp_cut(const PpParams& p, const PpGraph& g) {
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) {
    int drop = 0;
    if (p.alive[i]) {
      const unsigned u = (unsigned)p.a_u[i];
      const int v = p.a_v[i];
      drop = 1;
      for (int s = g.rowptr[v]; s < g.rowptr[v + 1]; ++s)
        if (g.csr[s] == u) { drop = 0; break; }          // alive (no kDead bit) and pointing back at u
    }
    p.mark[i] = (unsigned char)drop;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.a; i += blockDim.x)
    if (p.mark[i]) pp_kill(p, g, i);
  __syncthreads();
}

// utils.py:144-339 (the live branch): while some node has more than num_cameras-1 active out- (in-) edges, every
// such node drops its least probable active out- (in-) edge (lowest edge index on ties), all chosen on one snapshot
__device__ int pp_prune(const PpParams& p, const PpGraph& g) {
  int* flow_out = g.t0; int* flow_in = g.t1;
  unsigned long long* key_out = reinterpret_cast<unsigned long long*>(g.t2);   // t2|t3
  unsigned long long* key_in = reinterpret_cast<unsigned long long*>(g.t4);    // t4|t5
  int rounds = 0;
  for (;;) {
    for (int n = threadIdx.x; n < g.n; n += blockDim.x) {
      flow_out[n] = 0; flow_in[n] = 0; key_out[n] = ~0ull; key_in[n] = ~0ull;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.a; i += blockDim.x) {
      if (!p.alive[i]) continue;
      const int u = p.a_u[i], v = p.a_v[i];
      const unsigned long long key = ((unsigned long long)__float_as_uint(p.a_p[i]) << 32) | (unsigned)i;
      atomicAdd(&flow_out[u], 1);
      atomicAdd(&flow_in[v], 1);
      atomicMin(&key_out[u], key);
      atomicMin(&key_in[v], key);
    }
    __syncthreads();
    int any = 0;
    for (int n = threadIdx.x; n < g.n; n += blockDim.x) {
      if (flow_out[n] > g.cams - 1) { any = 1; pp_kill(p, g, (int)(key_out[n] & 0xffffffffu)); }
      if (flow_in[n] > g.cams - 1) { any = 1; pp_kill(p, g, (int)(key_in[n] & 0xffffffffu)); }
    }
    if (!__syncthreads_or(any)) return rounds;
    ++rounds;
  }
}

// Lane 0's part of utils.py:30-52: nx.strongly_connected_components' traversal (sources in node insertion order,
// successors in insertion order, a component is emitted when its root finishes -- Tarjan's algorithm; networkx
// evaluates the lowlinks at finish time, here they are folded into the single pass over the successors, which finds
// the same roots in the same order) and the stable placement of the emitted components by size.  Templated on the pointer types
// so that the LDS-resident case compiles to ds_read/ds_write instead of flat accesses.
template <typename IP, typename UP>
__device__ int pp_walk(IP rowptr, IP pre, IP low, IP comp, IP cursor, IP dstack, IP sstack, UP csr, IP src, int n_src) {
  int counter = 0, n_comp = 0, stop = 0;
  for (int i = 0; i < n_src; ++i) {
    const int source = src[i];
    if (pre[source] != 0) continue;
    int dtop = 0;
    dstack[dtop++] = source;
    pre[source] = low[source] = ++counter;
    sstack[stop++] = source;
    while (dtop) {
      const int v = dstack[dtop - 1];
      const int end = rowptr[v + 1];
      int c = cursor[v], lv = low[v];
      int child = -1;
      while (c < end) {
        const unsigned t = csr[c++];
        if (t & kDead) continue;
        const int pt = pre[t];
        if (pt == 0) { child = (int)t; break; }
        if (comp[t] < 0) lv = min(lv, pt);               // still on the component stack
      }
      cursor[v] = c;
      low[v] = lv;
      if (child >= 0) {
        dstack[dtop++] = child;
        pre[child] = low[child] = ++counter;
        sstack[stop++] = child;
        continue;
      }
      --dtop;
      if (dtop) {                                        // tree edge parent -> v: hand the lowlink up
        const int par = dstack[dtop - 1];
        if (lv < low[par]) low[par] = lv;
      }
      if (lv == pre[v]) {
        int w;
        do { w = sstack[--stop]; comp[w] = n_comp; } while (w != v);
        ++n_comp;
      }
    }
  }
  return n_comp;
}

template <typename IP>
__device__ void pp_place(IP clabel, IP hist, IP csize, int n_comp) {
  for (int k = 0; k < n_comp; ++k) clabel[k] = hist[csize[k] - 1]++;
}

typedef __attribute__((address_space(3))) int* LdsIntPtr;
typedef __attribute__((address_space(3))) unsigned* LdsUintPtr;

// utils.py:30-52.  Strongly connected components in the order networkx emits them, stably sorted by size; nodes
// without an active edge follow as singletons in index order.  label[v] = position of v's set; returns the number
// of sets; sizes by label in t5.  All lanes prepare the ordered source list and the numbering; the walk is lane 0's.
__device__ int pp_scc(const PpParams& p, const PpGraph& g, bool nodes_in_lds, bool csr_in_lds, int* sh) {
  int* pre = g.t0; int* low = g.t1; int* comp = g.t2; int* cursor = g.t3; int* dstack = g.t4; int* sstack = g.t5;
  int* src = g.t6;
  int* first_pos = low;                                  // until the walk starts
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) {
    pre[n] = 0; comp[n] = -1; cursor[n] = g.rowptr[n]; first_pos[n] = 0x7fffffff;
  }
  for (int i = threadIdx.x; i < 2 * g.a; i += blockDim.x) p.g_flags[i] = 0;
  __syncthreads();
  // node insertion order of nx.DiGraph(active edge list): first appearance, u before v within an edge
  for (int k = threadIdx.x; k < g.a; k += blockDim.x) {
    if (!p.alive[k]) continue;
    atomicMin(&first_pos[p.a_u[k]], 2 * k);
    atomicMin(&first_pos[p.a_v[k]], 2 * k + 1);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < g.n; n += blockDim.x)
    if (first_pos[n] != 0x7fffffff) p.g_flags[first_pos[n]] = 1;
  __syncthreads();
  const int n_src = block_exclusive_scan(p.g_flags, 2 * (int64_t)g.a, sh);
  for (int n = threadIdx.x; n < g.n; n += blockDim.x)
    if (first_pos[n] != 0x7fffffff) src[p.g_flags[first_pos[n]]] = n;
  __syncthreads();
  __shared__ int s_ncomp;
  if (threadIdx.x == 0) {
    const long long t_begin = wall_clock64();
    if (nodes_in_lds && csr_in_lds)
      s_ncomp = pp_walk((LdsIntPtr)g.rowptr, (LdsIntPtr)pre, (LdsIntPtr)low, (LdsIntPtr)comp, (LdsIntPtr)cursor,
                        (LdsIntPtr)dstack, (LdsIntPtr)sstack, (LdsUintPtr)g.csr, (LdsIntPtr)src, n_src);
    else
      s_ncomp = pp_walk(g.rowptr, pre, low, comp, cursor, dstack, sstack, g.csr, src, n_src);
    p.hdr[2] += (int)(wall_clock64() - t_begin);         // 100 MHz ticks spent walking (diagnostic, info[7])
  }
  __syncthreads();
  const int n_comp = s_ncomp;
  // sizes per emitted component (t3), histogram of sizes (t4, sizes 1..N), stable placement by size (lane 0),
  // isolated nodes numbered after them in index order (scan over t1)
  int* csize = g.t3; int* hist = g.t4; int* clabel = g.t0; int* lsize = g.t5; int* iso = g.t1;
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) { csize[n] = 0; hist[n] = 0; iso[n] = comp[n] < 0 ? 1 : 0; }
  __syncthreads();
  for (int n = threadIdx.x; n < g.n; n += blockDim.x)
    if (comp[n] >= 0) atomicAdd(&csize[comp[n]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < n_comp; k += blockDim.x) atomicAdd(&hist[csize[k] - 1], 1);
  __syncthreads();
  block_exclusive_scan(hist, g.n, sh);
  const int n_iso = block_exclusive_scan(iso, g.n, sh);
  if (threadIdx.x == 0) {
    if (nodes_in_lds) pp_place((LdsIntPtr)clabel, (LdsIntPtr)hist, (LdsIntPtr)csize, n_comp);
    else pp_place(clabel, hist, csize, n_comp);
  }
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) lsize[n] = 1;
  __syncthreads();
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) g.label[n] = comp[n] >= 0 ? clabel[comp[n]] : n_comp + iso[n];
  for (int k = threadIdx.x; k < n_comp; k += blockDim.x) lsize[clabel[k]] = csize[k];
  __syncthreads();
  return n_comp + n_iso;
}


// ------------------------------------------------------------------------------------------------
// Splitting loop without a full renumbering per iteration.  The loop only needs (a) the first over-sized set in the
// reference's numbering and (b) the set that carries a given number after an edge was dropped.  The numbering is
// "size, then networkx emission order"; emission order is "depth-first trees in order of their source's first
// appearance in the active edge list, components within a tree in finish order".  Dropping an edge changes trees
// only inside the weakly connected component (WCC) it belonged to, and positions in the edge list never move, so
// every component keeps the key (size, first position of its tree's source, sequence within its walk); lane 0
// re-walks the dirty WCCs alone and the numbers asked for are found by counting smaller keys.
// ------------------------------------------------------------------------------------------------
template <typename IP, typename UP>
__device__ void pp_walk_inc(IP rowptr, IP pre, IP low, IP comp, IP cursor, IP dstack, IP sstack, UP csr, IP src, int n_src,
                            IP csize, IP ctree, IP cseq, IP freel, int* n_free, int* n_ids) {
  int counter = 0, seq = 0, stop = 0, nf = *n_free, ni = *n_ids;
  for (int i = 0; i < n_src; ++i) {
    const int source = src[i];
    if (pre[source] != 0) continue;
    const int tree_key = low[source];                    // unvisited: still the node's first position in the edge list
    int dtop = 0;
    dstack[dtop++] = source;
    pre[source] = low[source] = ++counter;
    sstack[stop++] = source;
    while (dtop) {
      const int v = dstack[dtop - 1];
      const int end = rowptr[v + 1];
      int c = cursor[v], lv = low[v];
      int child = -1;
      while (c < end) {
        const unsigned t = csr[c++];
        if (t & kDead) continue;
        const int pt = pre[t];
        if (pt == 0) { child = (int)t; break; }
        if (comp[t] < 0) lv = min(lv, pt);
      }
      cursor[v] = c;
      low[v] = lv;
      if (child >= 0) {
        dstack[dtop++] = child;
        pre[child] = low[child] = ++counter;
        sstack[stop++] = child;
        continue;
      }
      --dtop;
      if (dtop) {
        const int par = dstack[dtop - 1];
        if (lv < low[par]) low[par] = lv;
      }
      if (lv == pre[v]) {
        const int id = nf > 0 ? freel[--nf] : ni++;
        int w, cnt = 0;
        do { w = sstack[--stop]; comp[w] = id; ++cnt; } while (w != v);
        csize[id] = cnt; ctree[id] = tree_key; cseq[id] = seq++;
      }
    }
  }
  *n_free = nf; *n_ids = ni;
}

__device__ void pp_wcc(const PpParams& p, const PpGraph& g) {
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) p.g_wcc[n] = n;
  __syncthreads();
  for (;;) {
    int changed = 0;
    for (int k = threadIdx.x; k < g.a; k += blockDim.x) {
      if (!p.alive[k]) continue;
      const int u = p.a_u[k], v = p.a_v[k];
      const int a = p.g_wcc[u], b = p.g_wcc[v];
      if (a < b) { atomicMin(&p.g_wcc[v], a); changed = 1; }
      else if (b < a) { atomicMin(&p.g_wcc[u], b); changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  // pointer jumping to the component minimum (labels only ever decrease towards it)
  for (;;) {
    int changed = 0;
    for (int n = threadIdx.x; n < g.n; n += blockDim.x) {
      const int w = p.g_wcc[n], ww = p.g_wcc[w];
      if (ww != w) { p.g_wcc[n] = ww; changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
}

// Re-walk the dirty WCCs.  s_cnt: {free ids, ids in use (high-water mark)} in shared memory.
__device__ void pp_inc_walk(const PpParams& p, const PpGraph& g, bool nodes_in_lds, bool csr_in_lds, int* s_cnt, int* sh) {
  int* pre = g.t0; int* low = g.t1; int* comp = g.t2; int* cursor = g.t3; int* dstack = g.t4; int* sstack = g.t5;
  int* src = g.t6;
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) {
    if (!p.g_dirty[p.g_wcc[n]]) continue;
    const int id = comp[n];
    if (id >= 0 && atomicExch(&g.csize[id], 0) > 0) g.freel[atomicAdd(&s_cnt[0], 1)] = id;
    pre[n] = 0; comp[n] = -1; cursor[n] = g.rowptr[n]; low[n] = 0x7fffffff;
  }
  for (int i = threadIdx.x; i < 2 * g.a; i += blockDim.x) p.g_flags[i] = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < g.a; k += blockDim.x) {
    if (!p.alive[k] || !p.g_dirty[p.g_wcc[p.a_u[k]]]) continue;
    atomicMin(&low[p.a_u[k]], 2 * k);
    atomicMin(&low[p.a_v[k]], 2 * k + 1);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < g.n; n += blockDim.x)
    if (p.g_dirty[p.g_wcc[n]] && low[n] != 0x7fffffff) p.g_flags[low[n]] = 1;
  __syncthreads();
  const int n_src = block_exclusive_scan(p.g_flags, 2 * (int64_t)g.a, sh);
  for (int n = threadIdx.x; n < g.n; n += blockDim.x)
    if (p.g_dirty[p.g_wcc[n]] && low[n] != 0x7fffffff) src[p.g_flags[low[n]]] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t_begin = wall_clock64();
    if (nodes_in_lds && csr_in_lds)
      pp_walk_inc((LdsIntPtr)g.rowptr, (LdsIntPtr)pre, (LdsIntPtr)low, (LdsIntPtr)comp, (LdsIntPtr)cursor,
                  (LdsIntPtr)dstack, (LdsIntPtr)sstack, (LdsUintPtr)g.csr, (LdsIntPtr)src, n_src,
                  (LdsIntPtr)g.csize, (LdsIntPtr)g.ctree, (LdsIntPtr)g.cseq, (LdsIntPtr)g.freel, &s_cnt[0], &s_cnt[1]);
    else
      pp_walk_inc(g.rowptr, pre, low, comp, cursor, dstack, sstack, g.csr, src, n_src, g.csize, g.ctree, g.cseq,
                  g.freel, &s_cnt[0], &s_cnt[1]);
    p.hdr[2] += (int)(wall_clock64() - t_begin);
  }
  __syncthreads();
}

// key(a) < key(b) in the reference's numbering: size, then tree source position, then sequence within the walk
__device__ __forceinline__ bool pp_key_less(const PpGraph& g, int a, int b) {
  if (g.csize[a] != g.csize[b]) return g.csize[a] < g.csize[b];
  if (g.ctree[a] != g.ctree[b]) return g.ctree[a] < g.ctree[b];
  return g.cseq[a] < g.cseq[b];
}

// Drop the dead edges from the active arrays (stable) once the cut / prune stages are over: the walks of the
// splitting loop then step over ~A_alive instead of A_in slots.  Dead edges get their prediction cleared here.
__device__ int pp_compact_alive(const PpParams& p, const PpGraph& g, int* sh) {
  int* pos = p.g_flags;
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) pos[i] = p.alive[i] ? 1 : 0;
  __syncthreads();
  const int kept = block_exclusive_scan(pos, g.a, sh);
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) {
    if (p.alive[i]) {
      const int k = pos[i];
      p.b_idx[k] = p.a_idx[i]; p.b_u[k] = p.a_u[i]; p.b_v[k] = p.a_v[i]; p.b_p[k] = p.a_p[i];
    } else {
      p.pred[p.a_idx[i]] = 0;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kept; k += blockDim.x) {
    p.a_idx[k] = p.b_idx[k]; p.a_u[k] = p.b_u[k]; p.a_v[k] = p.b_v[k]; p.a_p[k] = p.b_p[k];
    p.alive[k] = 1;
  }
  __syncthreads();
  return kept;
}

__global__ __launch_bounds__(kPpThreads) void pp_graph_kernel(PpParams p, int lds_nodes) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  __shared__ int sh[kPpThreads / 64 + 1];
  __shared__ unsigned s_min;
  if (p.hdr[1]) {                                        // capacity exceeded: report, leave predictions = argmax
    if (threadIdx.x == 0) p.info[3] = 1;
    return;
  }
  PpGraph g;
  g.n = (int)p.n_nodes; g.a = p.hdr[0]; g.cams = p.num_cameras;
  const size_t npad = ((size_t)g.n + 4) & ~(size_t)3;
  int* node_base = lds_nodes ? lds : p.g_node;
  g.rowptr = node_base; g.label = node_base + npad;
  g.t0 = node_base + 2 * npad; g.t1 = node_base + 3 * npad; g.t2 = node_base + 4 * npad; g.t3 = node_base + 5 * npad;
  g.t4 = node_base + 6 * npad; g.t5 = node_base + 7 * npad; g.t6 = node_base + 8 * npad;
  g.csize = node_base + 9 * npad; g.ctree = node_base + 10 * npad; g.cseq = node_base + 11 * npad;
  g.freel = node_base + 12 * npad;
  unsigned* lds_csr = reinterpret_cast<unsigned*>(lds + (lds_nodes ? kPpNodeArrays * npad : 0));
  bool csr_in_lds = g.a <= kPpLdsEdges;
  g.csr = csr_in_lds ? lds_csr : p.g_csr;

  pp_build_csr(p, g, sh);
  const bool cutting = p.flags & 1, pruning = p.flags & 2, splitting = p.flags & 4;
  int prune_rounds = 0, split_iters = 0, walks = 0, status = 0;
  if (cutting) pp_cut(p, g);
  if (pruning) prune_rounds = pp_prune(p, g);
  if (cutting) pp_cut(p, g);
  if (cutting || pruning) {
    g.a = pp_compact_alive(p, g, sh);
    csr_in_lds = g.a <= kPpLdsEdges;
    g.csr = csr_in_lds ? lds_csr : p.g_csr;
    pp_build_csr(p, g, sh);
  }
  if (splitting) {
    // utils.py:54-123, tail recursion unrolled: pick the first over-sized set of the numbering; drop every edge whose
    // probability equals the minimum over the active edges touching it; stay on the SAME NUMBER while the set that
    // carries it after the renumbering is over-sized (reference quirk), else pick again
    __shared__ int s_cnt[2], s_i[4];
    __shared__ unsigned long long s_key;
    int* over = g.t4;                                    // list of over-sized components (free outside the walks)
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    pp_wcc(p, g);
    for (int n = threadIdx.x; n < g.n; n += blockDim.x) { p.g_dirty[n] = 1; g.t2[n] = -1; g.csize[n] = 0; }
    __syncthreads();
    pp_inc_walk(p, g, lds_nodes != 0, csr_in_lds, s_cnt, sh);
    ++walks;
    for (;;) {
      // first over-sized set = smallest key among the components with more than num_cameras nodes
      if (threadIdx.x == 0) { s_i[0] = 0x7fffffff; s_key = ~0ull; s_i[1] = -1; s_i[2] = 0; }
      __syncthreads();
      const int n_ids = s_cnt[1];
      for (int c = threadIdx.x; c < n_ids; c += blockDim.x)
        if (g.csize[c] > g.cams) atomicMin(&s_i[0], g.csize[c]);
      __syncthreads();
      const int min_size = s_i[0];
      if (min_size == 0x7fffffff) break;
      for (int c = threadIdx.x; c < n_ids; c += blockDim.x)
        if (g.csize[c] == min_size)
          atomicMin(&s_key, ((unsigned long long)(unsigned)g.ctree[c] << 32) | (unsigned)g.cseq[c]);
      __syncthreads();
      for (int c = threadIdx.x; c < n_ids; c += blockDim.x)
        if (g.csize[c] == min_size && (((unsigned long long)(unsigned)g.ctree[c] << 32) | (unsigned)g.cseq[c]) == s_key)
          s_i[1] = c;
      __syncthreads();
      int cur = s_i[1];
      for (int c = threadIdx.x; c < n_ids; c += blockDim.x)      // its number = how many components precede it
        if (g.csize[c] > 0 && pp_key_less(g, c, cur)) atomicAdd(&s_i[2], 1);
      __syncthreads();
      const int lab = s_i[2];
      for (;;) {
        if (threadIdx.x == 0) s_min = 0xffffffffu;
        for (int n = threadIdx.x; n < g.n; n += blockDim.x) p.g_dirty[n] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < g.a; i += blockDim.x)
          if (p.alive[i] && (g.t2[p.a_u[i]] == cur || g.t2[p.a_v[i]] == cur))
            atomicMin(&s_min, __float_as_uint(p.a_p[i]));
        __syncthreads();
        const unsigned mn = s_min;
        if (mn == 0xffffffffu) { status = 2; break; }    // cannot happen for a real component; never spin on it
        for (int i = threadIdx.x; i < g.a; i += blockDim.x)
          if (p.alive[i] && __float_as_uint(p.a_p[i]) == mn) {
            pp_kill(p, g, i);
            p.g_dirty[p.g_wcc[p.a_u[i]]] = 1;
          }
        __syncthreads();
        pp_inc_walk(p, g, lds_nodes != 0, csr_in_lds, s_cnt, sh);
        ++walks; ++split_iters;
        if (split_iters >= kPpMaxSplitIters) { status = 3; break; }   // a resident workgroup must not run unbounded
        // which over-sized component carries number `lab` now?  (one wave per candidate counts the smaller keys)
        if (threadIdx.x == 0) { s_i[3] = 0; s_i[1] = -1; }
        __syncthreads();
        const int ids_now = s_cnt[1];
        for (int c = threadIdx.x; c < ids_now; c += blockDim.x)
          if (g.csize[c] > g.cams) over[atomicAdd(&s_i[3], 1)] = c;
        __syncthreads();
        const int n_over = s_i[3], lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
        for (int o = wave; o < n_over; o += n_waves) {
          const int id = over[o];
          int cnt = 0;
          for (int c = lane; c < ids_now; c += 64)
            if (g.csize[c] > 0 && pp_key_less(g, c, id)) ++cnt;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
          if (lane == 0 && cnt == lab) s_i[1] = id;
        }
        __syncthreads();
        cur = s_i[1];
        __syncthreads();
        if (cur < 0) break;
      }
      if (status) break;
    }
  }
  const int n_sets = pp_scc(p, g, lds_nodes != 0, csr_in_lds, sh);     // the reference's final numbering
  ++walks;
  // outputs
  int alive_cnt = 0;
  for (int i = threadIdx.x; i < g.a; i += blockDim.x) {
    if (p.alive[i]) ++alive_cnt; else p.pred[p.a_idx[i]] = 0;
  }
  for (int n = threadIdx.x; n < g.n; n += blockDim.x) p.id_pred[n] = g.label[n];
  __shared__ int s_alive;
  if (threadIdx.x == 0) s_alive = 0;
  __syncthreads();
  atomicAdd(&s_alive, alive_cnt);
  __syncthreads();
  if (threadIdx.x == 0) {
    p.info[1] = s_alive; p.info[2] = n_sets; p.info[3] = status ? status : (p.hdr[3] ? 4 : 0); p.info[4] = split_iters; p.info[5] = walks;
    p.info[6] = prune_rounds; p.info[7] = p.hdr[2];
  }
}

// ------------------------------------------------------------------------------------------------
struct PpLayout { size_t hdr, block_count, a_idx, a_u, a_v, a_p, a_slot, alive, mark, g_node, g_csr, g_flags, b_idx, b_u, b_v, b_p, g_wcc, g_dirty, total; };

static PpLayout pp_layout(int64_t n_nodes, int64_t n_edges, int64_t cap) {
  PpLayout lo;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
  const int64_t nb = (n_edges + kPpChunk - 1) / kPpChunk;
  lo.hdr = take(16 * sizeof(int));
  lo.block_count = take((size_t)(nb > 0 ? nb : 1) * sizeof(int));
  lo.a_idx = take((size_t)cap * 4); lo.a_u = take((size_t)cap * 4); lo.a_v = take((size_t)cap * 4);
  lo.a_p = take((size_t)cap * 4); lo.a_slot = take((size_t)cap * 4);
  lo.alive = take((size_t)cap); lo.mark = take((size_t)cap);
  lo.g_node = take((size_t)kPpNodeArrays * (n_nodes + 4) * 4);
  lo.g_csr = take((size_t)cap * 4);
  lo.g_flags = take((size_t)cap * 8);
  lo.b_idx = take((size_t)cap * 4); lo.b_u = take((size_t)cap * 4); lo.b_v = take((size_t)cap * 4); lo.b_p = take((size_t)cap * 4);
  lo.g_wcc = take((size_t)(n_nodes + 4) * 4); lo.g_dirty = take((size_t)(n_nodes + 4) * 4);
  lo.total = off;
  return lo;
}

size_t pp_workspace_bytes(int64_t n_nodes, int64_t n_edges, int64_t max_active) {
  const int64_t cap = (max_active > 0 && max_active < n_edges) ? max_active : n_edges;
  return pp_layout(n_nodes, n_edges, cap > 0 ? cap : 1).total;
}

int launch_postprocess(const float* logits, const int64_t* row, const int64_t* col, int64_t idx_stride, int64_t n_nodes,
                       int64_t n_edges, int num_cameras, int flags, int64_t max_active, float* prob1, int64_t* pred,
                       int64_t* id_pred, int32_t* info, void* workspace, size_t workspace_bytes, hipStream_t s) {
  int64_t cap = (max_active > 0 && max_active < n_edges) ? max_active : n_edges;
  if (cap < 1) cap = 1;
  const PpLayout lo = pp_layout(n_nodes, n_edges, cap);
  if (workspace_bytes < lo.total) return MTMC_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  PpParams p;
  p.logits = logits; p.row = row; p.col = col; p.idx_stride = idx_stride; p.n_nodes = n_nodes; p.n_edges = n_edges;
  p.num_cameras = num_cameras; p.flags = flags; p.prob1 = prob1; p.pred = pred; p.id_pred = id_pred; p.info = info;
  p.hdr = reinterpret_cast<int*>(ws + lo.hdr); p.block_count = reinterpret_cast<int*>(ws + lo.block_count);
  p.a_idx = reinterpret_cast<int*>(ws + lo.a_idx); p.a_u = reinterpret_cast<int*>(ws + lo.a_u);
  p.a_v = reinterpret_cast<int*>(ws + lo.a_v); p.a_p = reinterpret_cast<float*>(ws + lo.a_p);
  p.a_slot = reinterpret_cast<int*>(ws + lo.a_slot); p.alive = reinterpret_cast<unsigned char*>(ws + lo.alive);
  p.mark = reinterpret_cast<unsigned char*>(ws + lo.mark); p.g_node = reinterpret_cast<int*>(ws + lo.g_node);
  p.g_csr = reinterpret_cast<unsigned*>(ws + lo.g_csr); p.cap = cap;
  p.g_flags = reinterpret_cast<int*>(ws + lo.g_flags); p.b_idx = reinterpret_cast<int*>(ws + lo.b_idx);
  p.b_u = reinterpret_cast<int*>(ws + lo.b_u); p.b_v = reinterpret_cast<int*>(ws + lo.b_v);
  p.b_p = reinterpret_cast<float*>(ws + lo.b_p);
  p.g_wcc = reinterpret_cast<int*>(ws + lo.g_wcc); p.g_dirty = reinterpret_cast<int*>(ws + lo.g_dirty);
  const int nb = (int)((n_edges + kPpChunk - 1) / kPpChunk);
  if (nb > 0) hipLaunchKernelGGL(pp_classify_kernel, dim3(nb), dim3(256), 0, s, p);
  hipLaunchKernelGGL(pp_scan_kernel, dim3(1), dim3(1024), 0, s, p, nb);
  if (nb > 0) hipLaunchKernelGGL(pp_compact_kernel, dim3(nb), dim3(256), 0, s, p);
  const int lds_nodes = n_nodes <= kPpLdsNodes ? 1 : 0;
  const size_t npad = ((size_t)n_nodes + 4) & ~(size_t)3;
  const size_t lds = (lds_nodes ? kPpNodeArrays * npad * sizeof(int) : 0) + (size_t)kPpLdsEdges * sizeof(unsigned);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pp_graph_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (kPpNodeArrays * (kPpLdsNodes + 4) + kPpLdsEdges) * 4) != hipSuccess)
      return MTMC_E_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(pp_graph_kernel, dim3(1), dim3(kPpThreads), lds, s, p, lds_nodes);
  return hipGetLastError() == hipSuccess ? MTMC_OK : MTMC_E_HIP;
}

}  // namespace mtmc

extern "C" {

size_t mtmc_postprocess_workspace_bytes(int64_t n_nodes, int64_t n_edges, int64_t max_active) {
  if (n_nodes < 1 || n_edges < 0 || n_nodes >= (1ll << 31) || n_edges >= (1ll << 31)) return 0;
  return mtmc::pp_workspace_bytes(n_nodes, n_edges, max_active);
}

int32_t mtmc_postprocess(const float* logits, const int64_t* row, const int64_t* col, int64_t idx_stride,
                         int64_t n_nodes, int64_t n_edges, int32_t num_cameras, int32_t flags, int64_t max_active,
                         float* prob1, int64_t* predictions, int64_t* id_pred, int32_t* info,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (n_nodes < 1 || n_edges < 0 || n_nodes >= (1ll << 31) || n_edges >= (1ll << 31) || num_cameras < 1) return MTMC_E_ARG;
  if (!prob1 || !predictions || !id_pred || !info || !workspace || idx_stride < 1) return MTMC_E_ARG;
  if (n_edges > 0 && (!row || !col)) return MTMC_E_ARG;
  if (((uintptr_t)workspace & 255) || (logits && ((uintptr_t)logits & 7))) return MTMC_E_ARG;
  return mtmc::launch_postprocess(logits, row, col, idx_stride, n_nodes, n_edges, num_cameras, flags, max_active, prob1,
                                  predictions, id_pred, info, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

}  // extern "C"
