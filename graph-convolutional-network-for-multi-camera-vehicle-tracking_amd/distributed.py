"""Edge-range partitioned MPN forward over the GPUs of one node (not in the reference, which
is single-GPU).  One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI) moves the data.

Partitioning.  Rank r holds a contiguous slice of the edge list (`edge_index`, `edge_attr`, and therefore the
edge states and logits, which never leave the rank) and a contiguous slice of the node rows of `x` (the 8 KB/node
features stay local for the encoder GEMMs).  The 32-wide node state is replicated.

Exchanges per forward -- BatchNorm couples every row, so each statistics block is all-reduced (fp64, a few
hundred bytes to a few KB each), plus the two real data exchanges:
    after BEGIN + NODE_ENC/COMBINE 0      edge_attr moments + column statistics of encoder layer 0: ONE message
                                          (+ the degree, for mean aggregation only)
    after EDGE_ENC + NODE_ENC/COMBINE 1   hidden edge-encoder moments + column statistics of layer 1: ONE message
    after NODE_COMBINE l >= 2             column statistics of encoder layer l
    after NODE_H0    all-gather of the encoded node rows h0                       [N,32] f32
    per round        z1 statistics, then e' moments + z2 statistics together (two small all-reduces)
                     all-reduce (sum or max) of the aggregated node state h'      [N,32] f32
                     -- or, when every rank's edges have their source rows to themselves (row-sorted list, shards
                     snapped to row boundaries): no exchange of h' at all; each rank projects its own rows and the
                     ranks all-gather the column projections Pc                   [N,4]  f32   (8x fewer bytes),
                     and the final node state once at the end                     [N,32] f32
The degrees need no exchange: the node-update (z2) statistics are linear in every rank's share (node-only part over the
rows it projects with its local degrees, edge part over its edges), so only the 64 sums travel.

The phases are the single-GPU ones (mtmc_mpn_run_phases); this file decides their order (`step_plan`: the node
encoder's first layers run beside the edge branch so that their statistics share a message), what is exchanged behind
which of them, and issues everything between two collectives as ONE library call.  It talks to the kernels through the
small backend interface (prepare / run_phase_list / region / outputs / set_flags), which is what lets tests/ drive the
same code with a CPU stand-in over gloo.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib


def even_ranges(total: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) ranges, the first `total % parts` one element longer."""
    base, extra = divmod(total, parts)
    out, lo = [], 0
    for r in range(parts):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def edge_ranges(row: Optional[torch.Tensor], n_edges: int, parts: int, snap_to_rows: bool = False):
    """Edge ranges per rank; with `snap_to_rows` (row-sorted lists) boundaries move to the next row change so
    that every node's out-edges live on one rank."""
    ranges = even_ranges(n_edges, parts)
    if not snap_to_rows or row is None or n_edges == 0:
        return ranges
    cuts = [0]
    for lo, hi in ranges[:-1]:
        c = max(hi, cuts[-1])
        if 0 < c < n_edges:
            same = (row[c:] != row[c - 1]).nonzero()
            c = n_edges if same.numel() == 0 else c + int(same[0])
        cuts.append(min(c, n_edges))
    cuts.append(n_edges)
    return [(cuts[i], cuts[i + 1]) for i in range(parts)]


def tile_rows(row_ranges, n_nodes):
    """Row-disjoint source-row ranges (row_ranges_of) stretched to a partition of [0, N): rows without out-edges between
    two ranges go to the earlier one, so that every node row is projected, cleared and reported by exactly one rank."""
    world = len(row_ranges)
    order = sorted((r for r in range(world) if row_ranges[r][1] > row_ranges[r][0]), key=lambda r: row_ranges[r][0])
    tiled = [(0, 0)] * world
    for i, r in enumerate(order):
        lo = 0 if i == 0 else row_ranges[r][0]
        hi = n_nodes if i == len(order) - 1 else row_ranges[order[i + 1]][0]
        tiled[r] = (lo, hi)
    if not order:
        tiled[0] = (0, n_nodes)
    return tiled


def step_plan(spec, local_rows: bool, own_rows: bool):
    """One forward as [(phases, (exchange, index))]: `phases` = the (phase, arg) pairs that run back to back in one
    library call, `exchange` = what travels behind them (None: nothing).  The statistics all-reduces sit at the dependency
    points only, and the edge branch's two ride with the node encoder's first two (their blocks are adjacent in the
    workspace): L = 3 on row-complete shards = 4 + 3 * 3 collectives + the final node state, 14 in all -- against one
    library call and one collective per phase before (about 30).  Dependencies: EDGE_ENC needs the reduced edge_attr moments
    (BEGIN), encoder layer l + 1 the reduced column statistics of layer l, round 0 the reduced hidden edge-encoder moments;
    inside a round pass A needs all of Pc, pass B the z1 statistics, pass C the e'-moment + z2 statistics; on row-complete
    shards pass C of round r and the projection of round r + 1 need nothing from other ranks in between."""
    P = _lib
    n_layers, L = len(spec.enc_node), spec.num_enc_steps
    steps = [([(P.PH_BEGIN, 0), (P.PH_NODE_ENC, 0), (P.PH_NODE_COMBINE, 0)], ("enc_merged", 0))]
    if n_layers > 1:
        steps.append(([(P.PH_EDGE_ENC, 0), (P.PH_NODE_ENC, 1), (P.PH_NODE_COMBINE, 1)], ("enc_merged", 1)))
    else:
        steps.append(([(P.PH_EDGE_ENC, 0)], ("stat_enc2", 0)))
    for l in range(2, n_layers):
        steps.append(([(P.PH_NODE_ENC, l), (P.PH_NODE_COMBINE, l)], ("stat_enc_node", l)))
    pending = [(P.PH_NODE_H0, 0)]
    if not own_rows:
        steps.append((pending, ("h0", 0)))
        pending = []
    for r in range(L):
        pending.append((P.PH_ROUND_PROJ, r))
        if local_rows:
            steps.append((pending, ("Pc", r)))
            pending = []
        pending.append((P.PH_ROUND_A, r))
        steps.append((pending, ("round_z1", r)))
        steps.append(([(P.PH_ROUND_B, r), (P.PH_ROUND_STAT, r)], ("round_m_z2", r)))
        pending = [(P.PH_ROUND_C, r)]
        if not local_rows:
            steps.append((pending, ("agg", r)))
            pending = []
    pending.append((P.PH_END, 0))
    steps.append((pending, (None, 0)))
    return steps


class ShardedForward:
    """Runs one forward on this rank's shard.  `backend` is a ForwardEngine (HIP) or anything with the same
    five methods."""

    def __init__(self, backend, spec, group=None, exchange_alone=False):
        """exchange_alone: run the row exchanges also in a group of ONE rank (where they move nothing) -- a rehearsal of the
        RCCL calls on a single-GPU box (tests/test_gpu_sharded_forward.py); off by default."""
        self.backend, self.spec, self.group, self.exchange_alone = backend, spec, group, exchange_alone

    # collectives (in place on workspace views)
    def _sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _max(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def _gather_rows(self, h, row_ranges, rank, world):
        """Every rank holds COMPLETE rows [lo_r, hi_r) of h (row-snapped shards of a row-sorted list): exchange them
        with one all-gather (half the bytes of the all-reduce they make unnecessary), padded to the longest range."""
        longest = max(hi - lo for lo, hi in row_ranges)
        if longest == 0 or (world == 1 and not self.exchange_alone):
            return
        lo, hi = row_ranges[rank]
        send = h.new_empty((longest, h.shape[1]))      # the padding rows are never copied out
        send[:hi - lo] = h[lo:hi]
        if dist.get_backend(self.group) == "nccl":
            recv = h.new_empty((world * longest, h.shape[1]))
            dist.all_gather_into_tensor(recv, send, group=self.group)
            parts = recv.view(world, longest, h.shape[1])
        else:                                      # gloo (tests): no all-gather of device tensors, stage on the host
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=self.group)
        for r, (rlo, rhi) in enumerate(row_ranges):
            if r != rank and rhi > rlo:
                h[rlo:rhi] = parts[r][:rhi - rlo].to(h.device)

    def __call__(self, x_local, node_range, edge_index_local, edge_attr_local, n_edges_total, row_ranges=None,
                 own_rows=False, replicate_h=True):
        """x_local: rows [node_range[0], node_range[1]) of x; node_range = (lo, hi, N).
        edge_index_local / edge_attr_local: this rank's edge slice (global node ids).
        row_ranges: per rank, the node range [lo, hi) that contains ALL source rows of its edge slice and no source
        row of any other rank's (see `row_ranges_of`).  When given, a rank aggregates complete rows, so it keeps the node
        state of its own rows to itself: it projects only those rows (Pr, Q are only ever read at an edge's source row)
        and the ranks exchange the column projections Pc -- 16 bytes per node and round instead of the 128 bytes of the
        node state -- plus the final node state once, for the replicated output.  Without it the aggregated state is
        all-reduced every round and every rank projects every node.
        own_rows (needs row_ranges): the rank ENCODES the node rows it projects -- node_range[:2] is its entry of
        `tile_rows(row_ranges, N)` instead of the even split -- so the encoded state h0 needs no exchange either.
        replicate_h=False (with row_ranges): skip the final all-gather; the returned node state is then valid only in the
        rank's own rows `tile_rows(row_ranges, N)[rank]` (the callers of the reference never read it: `outputs, _ = ...`)."""
        be, spec = self.backend, self.spec
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        local_rows = row_ranges is not None
        if local_rows:
            row_ranges = tile_rows(row_ranges, node_range[2])
        prep = be.prepare(x_local, edge_index_local, edge_attr_local, n_edges_total=n_edges_total,
                          node_range=node_range, **({"row_range": row_ranges[rank]} if local_rows else {}))
        n = node_range[2]
        mean = spec.agg == "mean"
        if mean:
            be.set_flags(prep, _lib.F_GLOBAL_DEG)        # (replaces the flags: the sharded path sums with atomics)
        if own_rows and not local_rows:
            raise ValueError("own_rows needs row_ranges (row-complete edge shards)")
        rows = row_ranges if own_rows else even_ranges(n, world)
        if tuple(rows[rank]) != tuple(node_range[:2]):
            raise ValueError(f"rank {rank} must encode node rows {rows[rank]} "
                             f"({'tile_rows(row_ranges, N)' if own_rows else 'even_ranges'}), got {node_range[:2]}")

        for phases, (what, idx) in step_plan(spec, local_rows, own_rows):
            be.run_phase_list(prep, phases)             # ONE library call for everything between two collectives
            if what == "enc_merged":                    # edge-branch block + encoder layer idx's column statistics: one message
                for t in be.region(prep, "enc_merged", idx):
                    self._sum(t)
                if idx == 0 and mean:                   # (the degree is complete after BEGIN)
                    g = be.region(prep, "deg_global")
                    g.copy_(be.region(prep, "deg"))
                    if not local_rows:                  # row-complete shards: a row's whole degree is already local
                        self._sum(g)
            elif what in ("stat_enc2", "stat_enc_node", "round_z1"):
                self._sum(be.region(prep, what, idx))
            elif what == "round_m_z2":                  # pass B's e' moments and the node-update (z2) sums -- node-only part
                for t in be.region(prep, "round_m_z2", idx):   # from ROUND_PROJ, edge part from ROUND_B -- are adjacent: one message
                    self._sum(t)
            elif what == "h0":
                h0 = be.region(prep, "h0")
                even = all(hi - lo == rows[0][1] - rows[0][0] for lo, hi in rows)
                if even and dist.get_backend(self.group) == "nccl":
                    dist.all_gather_into_tensor(h0, h0[rows[rank][0]:rows[rank][1]].clone(), group=self.group)
                else:                                   # uneven split: one broadcast per owner
                    for r, (lo, hi) in enumerate(rows):
                        if hi > lo:
                            dist.broadcast(h0[lo:hi], src=dist.get_global_rank(self.group, r) if self.group else r,
                                           group=self.group)
            elif what == "Pc":                          # every rank gathers along its edges' columns: all of Pc
                self._gather_rows(be.region(prep, "Pc"), row_ranges, rank, world)
            elif what == "agg":
                (self._max if spec.agg == "max" else self._sum)(be.region(prep, "agg", idx))
        logits, h = be.outputs(prep)
        if local_rows and replicate_h and (spec.num_enc_steps > 0 or own_rows):   # every rank's rows of the final state
            self._gather_rows(h, row_ranges, rank, world)
        return logits, h


def row_ranges_of(edge_index_local, group=None):
    """[lo, hi) of the source rows in every rank's edge slice, or None if the slices are not row-disjoint (then the
    node states must be all-reduced).  One tiny all-gather; call it once per graph, not per forward."""
    world = dist.get_world_size(group)
    row = edge_index_local[0]
    mine = torch.tensor([int(row.min()), int(row.max()) + 1] if row.numel() else [0, 0], dtype=torch.int64, device=row.device)
    if dist.get_backend(group) == "nccl":
        allr = torch.empty(2 * world, dtype=torch.int64, device=row.device)
        dist.all_gather_into_tensor(allr, mine, group=group)
        allr = allr.view(world, 2).tolist()
    else:
        parts = [torch.empty(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, mine.cpu(), group=group)
        allr = [p.tolist() for p in parts]
    ranges = [(int(a), int(b)) for a, b in allr]
    live = sorted(r for r in ranges if r[1] > r[0])
    if any(a[1] > b[0] for a, b in zip(live, live[1:])):
        return None
    return ranges


def sharded_forward(module, x_local, node_range, edge_index_local, edge_attr_local, n_edges_total, group=None,
                    row_ranges=None, own_rows=False, replicate_h=True, exchange_alone=False):
    """Convenience wrapper: one edge-partitioned forward of a (HIP-backed) MOTMPNet on this rank's shard.
    Returns ({'classified_edges': [local logits]}, h) with h replicated on every rank (unless replicate_h=False)."""
    from . import engine
    if module._engine is None:
        module._engine = engine.ForwardEngine(module)
    logits, h = ShardedForward(module._engine, module.spec, group, exchange_alone)(x_local, node_range, edge_index_local,
                                                                  edge_attr_local, n_edges_total, row_ranges, own_rows,
                                                                  replicate_h)
    return {"classified_edges": logits}, h
