"""CPU stand-in for the HIP phase backend -- TEST INFRASTRUCTURE.

Implements the five-method backend interface of mtmc_mpn.distributed.ShardedForward
(prepare / run_phase_list / region / outputs / set_flags) with torch CPU ops, phase by phase, using the same
workspace *regions* (names, shapes, replica layout) as csrc/api.hip.  The world_size-2 gloo test drives the
real orchestration code (which collective on which region after which phase) through it.  The per-phase math
is tests/phase_model.py's, cut at the same points as the kernels.  Never imported by the product.
"""
import types

import torch

from mtmc_mpn import _lib
from phase_model import F64, PhaseModel, bn_affine

R = _lib.STAT_REPLICAS


def tri(n, i, j):
    return i * n - i * (i - 1) // 2 + (j - i)


def pack_moments(v64):
    """m1 | packed upper triangle of sum v v^T, as the kernels store it."""
    d = v64.shape[1]
    m2 = v64.t() @ v64
    return torch.cat([v64.sum(0), torch.stack([m2[i, j] for i in range(d) for j in range(i, d)])])


def unpack_moments(flat, d):
    m1 = flat[:d]
    m2 = torch.zeros(d, d, dtype=F64)
    for i in range(d):
        for j in range(i, d):
            m2[i, j] = m2[j, i] = flat[d + tri(d, i, j)]
    return m1, m2


def moments_affine(w, b, m1, m2, count, gamma, beta):
    w, b = w.to(F64), b.to(F64)
    wm1 = w @ m1
    s = wm1 + b * count
    q = ((w @ m2) * w).sum(1) + 2 * b * wm1 + b * b * count
    return bn_affine(s, q, count, gamma, beta)


class CpuPhaseBackend:
    def __init__(self, sd, spec):
        self.sd, self.spec = sd, spec
        self.pm = PhaseModel(sd, spec)

    def p(self, name):
        return self.sd[name]

    def phase_list(self):
        s = self.spec
        seq = [(_lib.PH_BEGIN, 0), (_lib.PH_EDGE_ENC, 0)]
        for l in range(len(s.enc_node)):
            seq += [(_lib.PH_NODE_ENC, l), (_lib.PH_NODE_COMBINE, l)]
        seq += [(_lib.PH_NODE_H0, 0)]
        for r in range(s.num_enc_steps):
            seq += [(_lib.PH_ROUND_PROJ, r), (_lib.PH_ROUND_A, r), (_lib.PH_ROUND_B, r), (_lib.PH_ROUND_STAT, r),
                    (_lib.PH_ROUND_C, r)]
        return seq + [(_lib.PH_END, 0)]

    def prepare(self, x, edge_index, edge_attr, training=False, n_edges_total=None, node_range=None, row_range=None):
        s = self.spec
        lo, hi, n = node_range if node_range is not None else (0, x.shape[0], x.shape[0])
        e = edge_index.shape[1]
        L = s.num_enc_steps
        n_out = min(s.num_class_steps, L) if L > 0 else 1
        c = types.SimpleNamespace(
            x=x, row=edge_index[0], col=edge_index[1], attr=edge_attr, n=n, e=e, lo=lo, hi=hi,
            e_total=float(e if n_edges_total is None else n_edges_total), flags=0,
            stat_attr=torch.zeros(R * _lib.ATTR_STRIDE, dtype=F64), stat_enc2=torch.zeros(R * _lib.ENC2_STRIDE, dtype=F64),
            stat_enc_node=[torch.zeros(2 * l.out_dim, dtype=F64) for l in s.enc_node],
            round_z1=[torch.zeros(R * _lib.Z1_STRIDE, dtype=F64) for _ in range(L)],
            round_m=[torch.zeros(R * _lib.M_STRIDE, dtype=F64) for _ in range(L)],
            round_z2=[torch.zeros(R * _lib.Z2_STRIDE, dtype=F64) for _ in range(L)],
            deg=torch.zeros(n, dtype=torch.int32), deg_global=torch.zeros(n, dtype=torch.int32),
            seg=torch.zeros(n, 4, dtype=F64), h0=torch.zeros(n, 32), h_acc=[torch.zeros(n, 32), torch.zeros(n, 32)],
            h=torch.zeros(n, 32), logits=torch.zeros(n_out, e, s.cls_edge[0].out_dim), n_out=n_out,
            Y=[None] * len(s.enc_node), e_buf=None, P=torch.full((n, 8), float("nan")), Q=None, enc_aff=None,
            rows=(0, n) if row_range is None else (int(row_range[0]), int(row_range[1])))
        return c

    def set_flags(self, c, flags):
        c.flags = flags

    def run_phase_list(self, c, pairs):
        for ph, arg in pairs:
            self.run_phase(c, ph, arg)

    def region(self, c, name, idx=0):
        if name in ("stat_attr", "stat_enc2", "deg", "deg_global", "h0"):
            return getattr(c, name)
        if name in ("stat_enc_node", "round_z1", "round_m", "round_z2"):
            return getattr(c, name)[idx]
        if name == "enc_merged":                   # (adjacent blocks in the HIP workspace: one message there, two tensors here)
            return (c.stat_attr if idx == 0 else c.stat_enc2, c.stat_enc_node[idx])
        if name == "round_m_z2":
            return (c.round_m[idx], c.round_z2[idx])
        if name == "Pc":
            return c.P[:, 4:]
        if name == "agg":
            return c.h if (idx == self.spec.num_enc_steps - 1 and self.spec.agg != "mean") else c.h_acc[idx & 1]
        raise KeyError(name)

    def outputs(self, c):
        return [c.logits[i] for i in range(c.n_out)], c.h

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _rep_sum(block, stride, count):
        return block.view(R, stride).sum(0)[:count]

    def _enc_affines(self, c):
        l1, l2 = self.spec.enc_edge
        fe = l1.in_dim
        m1, m2 = unpack_moments(self._rep_sum(c.stat_attr, _lib.ATTR_STRIDE, fe + fe * (fe + 1) // 2), fe)
        s1, t1 = moments_affine(self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.weight"),
                                self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.bias"), m1, m2, c.e_total,
                                self.p(f"encoder.edge_mlp.fc_layers.{l1.bn_slot}.weight"),
                                self.p(f"encoder.edge_mlp.fc_layers.{l1.bn_slot}.bias"))
        return s1, t1

    def _enc_affines2(self, c):
        s1, t1 = self._enc_affines(c)
        l2 = self.spec.enc_edge[1]
        m1, m2 = unpack_moments(self._rep_sum(c.stat_enc2, _lib.ENC2_STRIDE, 14), 4)
        s2, t2 = moments_affine(self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.weight"),
                                self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.bias"), m1, m2, c.e_total,
                                self.p(f"encoder.edge_mlp.fc_layers.{l2.bn_slot}.weight"),
                                self.p(f"encoder.edge_mlp.fc_layers.{l2.bn_slot}.bias"))
        return (s1, t1, s2, t2)

    def _weights(self):
        s = self.spec
        le, ln = s.upd_edge[0], s.upd_node[0]
        pre_e, pre_n = "MPNet.edge_model.edge_mlp.fc_layers.", "MPNet.node_model.node_mlp.fc_layers."
        we, wn = self.p(f"{pre_e}{le.lin_slot}.weight"), self.p(f"{pre_n}{ln.lin_slot}.weight")
        hn = (2 if s.reattach_nodes else 1) * 32
        return types.SimpleNamespace(
            w_pr=we[:, :hn], w_pc=we[:, hn:2 * hn], w_ee=we[:, 2 * hn:], be=self.p(f"{pre_e}{le.lin_slot}.bias"),
            ge=self.p(f"{pre_e}{le.bn_slot}.weight"), bte=self.p(f"{pre_e}{le.bn_slot}.bias"),
            w_q=wn[:, :hn], w_a=wn[:, hn:], bn=self.p(f"{pre_n}{ln.lin_slot}.bias"),
            gn=self.p(f"{pre_n}{ln.bn_slot}.weight"), btn=self.p(f"{pre_n}{ln.bn_slot}.bias"))

    def _z1(self, c, r):
        s, w = self.spec, self._weights()
        e0 = self.pm.e0(c.attr, c.enc_aff) if (r == 0 or s.reattach_edges) else None
        e_prev = e0 if r == 0 else c.e_buf
        e_in = torch.cat([e0, e_prev], 1) if s.reattach_edges else e_prev
        return c.P[c.row, :4] + c.P[c.col, 4:] + e_in @ w.w_ee.t() + w.be

    # -- phases -------------------------------------------------------------------------------------
    def run_phase(self, c, ph, arg):
        s = self.spec
        if ph == _lib.PH_BEGIN:
            c.stat_attr.zero_()
            c.stat_attr[:0 + len(pack_moments(c.attr.to(F64)))] = pack_moments(c.attr.to(F64))
            c.deg.zero_()
            c.deg.index_add_(0, c.row, torch.ones(c.e, dtype=torch.int32))
        elif ph == _lib.PH_EDGE_ENC:
            s1, t1 = self._enc_affines(c)
            u = self.pm._enc1(c.attr, s1, t1)
            c.stat_enc2.zero_()
            c.stat_enc2[:14] = pack_moments(u.to(F64))
        elif ph == _lib.PH_NODE_ENC:
            lay = s.enc_node[arg]
            if arg == 0:
                a = c.x
            else:
                prev = s.enc_node[arg - 1]
                st = c.stat_enc_node[arg - 1]
                sc, sh = bn_affine(st[:prev.out_dim], st[prev.out_dim:], float(c.n),
                                   self.p(f"encoder.node_mlp.fc_layers.{prev.bn_slot}.weight"),
                                   self.p(f"encoder.node_mlp.fc_layers.{prev.bn_slot}.bias"))
                a = torch.relu(c.Y[arg - 1] * sc + sh)
            y = a @ self.p(f"encoder.node_mlp.fc_layers.{lay.lin_slot}.weight").t() + \
                self.p(f"encoder.node_mlp.fc_layers.{lay.lin_slot}.bias")
            c.Y[arg] = y
            c.stat_enc_node[arg][:] = torch.cat([y.to(F64).sum(0), (y.to(F64) ** 2).sum(0)])
        elif ph == _lib.PH_NODE_COMBINE:
            pass
        elif ph == _lib.PH_NODE_H0:
            lay = s.enc_node[-1]
            st = c.stat_enc_node[-1]
            sc, sh = bn_affine(st[:32], st[32:], float(c.n), self.p(f"encoder.node_mlp.fc_layers.{lay.bn_slot}.weight"),
                               self.p(f"encoder.node_mlp.fc_layers.{lay.bn_slot}.bias"))
            c.h0.fill_(float("nan"))                # rows this rank does not encode: poison until exchanged
            c.h0[c.lo:c.hi] = torch.relu(c.Y[-1] * sc + sh)
        elif ph == _lib.PH_ROUND_PROJ:
            w = self._weights()
            if arg == 0:
                c.enc_aff = self._enc_affines2(c)
                h = c.h0
            else:
                h = c.h_acc[(arg - 1) & 1]
                if s.agg == "mean":
                    deg = c.deg_global if (c.flags & _lib.F_GLOBAL_DEG) else c.deg
                    h = h / deg.clamp(min=1).float()[:, None]
            hcat = torch.cat([c.h0, h], 1) if s.reattach_nodes else h
            lo, hi = c.rows                        # like the kernels: only these rows are projected and cleared; the
            c.P.fill_(float("nan"))                # rest is poison until the host has exchanged it
            c.P[lo:hi] = torch.cat([hcat[lo:hi] @ w.w_pr.t(), hcat[lo:hi] @ w.w_pc.t()], 1)
            c.Q = torch.full((c.n, 32), float("nan"))
            c.Q[lo:hi] = hcat[lo:hi] @ w.w_q.t()
            tgt = self.region(c, "agg", arg)
            if (lo, hi) != (0, c.n):
                tgt.fill_(float("nan"))
            tgt[lo:hi] = 0
        elif ph == _lib.PH_ROUND_A:
            z1 = self._z1(c, arg).to(F64)
            c.round_z1[arg].zero_()
            c.round_z1[arg][:8] = torch.cat([z1.sum(0), (z1 ** 2).sum(0)])
        elif ph == _lib.PH_ROUND_B:
            w = self._weights()
            st = self._rep_sum(c.round_z1[arg], _lib.Z1_STRIDE, 8)
            s1, t1 = bn_affine(st[:4], st[4:], c.e_total, w.ge, w.bte)
            e_new = torch.relu(self._z1(c, arg) * s1 + t1)
            c.e_buf = e_new
            c.round_m[arg].zero_()
            c.round_m[arg][:14] = pack_moments(e_new.to(F64))
            c.seg.index_add_(0, c.row, e_new.to(F64))
        elif ph == _lib.PH_ROUND_STAT:
            w = self._weights()
            lo, hi = c.rows
            qb = (c.Q[lo:hi] + w.bn).to(F64)
            proj = c.seg[lo:hi] @ w.w_a.to(F64).t()
            deg = c.deg[lo:hi].to(F64)[:, None]
            c.round_z2[arg].zero_()
            c.round_z2[arg][:64] = torch.cat([(deg * qb + proj).sum(0), (deg * qb * qb + 2 * qb * proj).sum(0)])
            c.seg.zero_()
        elif ph == _lib.PH_ROUND_C:
            w = self._weights()
            _, m2 = unpack_moments(self._rep_sum(c.round_m[arg], _lib.M_STRIDE, 14), 4)
            st = self._rep_sum(c.round_z2[arg], _lib.Z2_STRIDE, 64)
            a64 = w.w_a.to(F64)
            s2, t2 = bn_affine(st[:32], st[32:] + ((a64 @ m2) * a64).sum(1), c.e_total, w.gn, w.btn)
            m = torch.relu((c.Q[c.row] + c.e_buf @ w.w_a.t() + w.bn) * s2 + t2)
            tgt = self.region(c, "agg", arg)
            idx = c.row.view(-1, 1).expand_as(m)
            if s.agg == "max":
                tgt.copy_(tgt.scatter_reduce(0, idx, m, reduce="amax", include_self=True))
            else:
                tgt.copy_(tgt.double().index_add_(0, c.row, m.double()).float())
            step, first = arg + 1, max(1, s.num_enc_steps - s.num_class_steps + 1)
            if step >= first:
                c.logits[step - first] = self.pm.classify(c.e_buf)
        elif ph == _lib.PH_END:
            L = s.num_enc_steps
            if L == 0:
                c.h.copy_(c.h0)
                c.logits[0] = self.pm.classify(self.pm.e0(c.attr, self._enc_affines2(c)))
            elif s.agg == "mean":
                deg = c.deg_global if (c.flags & _lib.F_GLOBAL_DEG) else c.deg
                c.h.copy_(c.h_acc[(L - 1) & 1] / deg.clamp(min=1).float()[:, None])
        else:
            raise ValueError(ph)
