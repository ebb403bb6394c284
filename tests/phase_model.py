"""Executable specification of the kernel decomposition -- TEST INFRASTRUCTURE.

A torch/CPU model of exactly the algebra the HIP kernels implement, phase by
phase, with the same precision policy (fp32 element math, fp64 statistics):

  * node->edge "gather" through per-node projections:  W.[h[row] | h[col] | e] = Pr[row] + Pc[col] + We_e.e
  * BatchNorm batch statistics from fp64 moments (sum, sum of squares / second-moment matrices)
    instead of materialising the pre-activations:
      - edge encoder layer 1 from the moments of edge_attr, layer 2 from the moments of its input
      - node-update layer (32 wide, over E rows) from per-node segment sums S_i = sum_{row=i} e',
        the degree, and the global 4x4 second-moment matrix of e'
  * e0 = edge_encoder(edge_attr) is never stored: it is recomputed from edge_attr where needed.

It takes the per-rank `reduce` hook the multi-GPU path uses (an all-reduce over edge shards), so the
world_size-2 gloo test can run the *same* phase sequence the GPU ranks run.  It never imports the
oracle and is never imported by the product.
"""
from __future__ import annotations

import torch

EPS = 1e-5
F64 = torch.float64


def bn_affine(sum_, sumsq, count, gamma, beta):
    """(sum, sum of squares) in fp64 -> per-channel scale/shift in fp32: y = s*z + t."""
    mean = sum_ / count
    var = (sumsq / count - mean * mean).clamp_(min=0)
    s = gamma.to(F64) / torch.sqrt(var + EPS)
    t = beta.to(F64) - mean * s
    return s.float(), t.float()


class PhaseModel:
    def __init__(self, sd, spec, reduce=None):
        self.sd, self.spec = sd, spec
        self.reduce = reduce if reduce is not None else (lambda t: t)

    def p(self, name):
        return self.sd[name]

    # ---- node encoder: GEMM chain, BN stats fused as column sums --------------------------
    def encode_nodes(self, x, n_total=None):
        a = x
        n_total = x.shape[0] if n_total is None else n_total
        for layer in self.spec.enc_node:
            w = self.p(f"encoder.node_mlp.fc_layers.{layer.lin_slot}.weight")
            b = self.p(f"encoder.node_mlp.fc_layers.{layer.lin_slot}.bias")
            y = a @ w.t() + b
            st = self.reduce(torch.stack([y.to(F64).sum(0), (y.to(F64) ** 2).sum(0)]))
            s, t = bn_affine(st[0], st[1], n_total,
                             self.p(f"encoder.node_mlp.fc_layers.{layer.bn_slot}.weight"),
                             self.p(f"encoder.node_mlp.fc_layers.{layer.bn_slot}.bias"))
            a = torch.relu(y * s + t)
        return a

    # ---- edge encoder by moments ----------------------------------------------------------
    def edge_encoder_affines(self, attr, e_total):
        l1, l2 = self.spec.enc_edge
        w1 = self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.weight").to(F64)
        b1 = self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.bias").to(F64)
        a64 = attr.to(F64)
        mom = self.reduce(torch.cat([a64.sum(0), (a64.t() @ a64).reshape(-1)]))
        fe = attr.shape[1]
        m1, m2 = mom[:fe] / e_total, mom[fe:].reshape(fe, fe) / e_total
        mean1 = w1 @ m1 + b1
        ez2 = ((w1 @ m2) * w1).sum(1) + 2 * b1 * (w1 @ m1) + b1 * b1
        s1, t1 = bn_affine(mean1 * e_total, ez2 * e_total, e_total,
                           self.p(f"encoder.edge_mlp.fc_layers.{l1.bn_slot}.weight"),
                           self.p(f"encoder.edge_mlp.fc_layers.{l1.bn_slot}.bias"))
        u = self._enc1(attr, s1, t1)
        u64 = u.to(F64)
        mom = self.reduce(torch.cat([u64.sum(0), (u64.t() @ u64).reshape(-1)]))
        w2 = self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.weight").to(F64)
        b2 = self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.bias").to(F64)
        d = u.shape[1]
        m1, m2 = mom[:d] / e_total, mom[d:].reshape(d, d) / e_total
        mean2 = w2 @ m1 + b2
        ez2 = ((w2 @ m2) * w2).sum(1) + 2 * b2 * (w2 @ m1) + b2 * b2
        s2, t2 = bn_affine(mean2 * e_total, ez2 * e_total, e_total,
                           self.p(f"encoder.edge_mlp.fc_layers.{l2.bn_slot}.weight"),
                           self.p(f"encoder.edge_mlp.fc_layers.{l2.bn_slot}.bias"))
        return (s1, t1, s2, t2)

    def _enc1(self, attr, s1, t1):
        l1 = self.spec.enc_edge[0]
        w1 = self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.weight")
        b1 = self.p(f"encoder.edge_mlp.fc_layers.{l1.lin_slot}.bias")
        return torch.relu((attr @ w1.t() + b1) * s1 + t1)

    def e0(self, attr, aff):
        s1, t1, s2, t2 = aff
        l2 = self.spec.enc_edge[1]
        w2 = self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.weight")
        b2 = self.p(f"encoder.edge_mlp.fc_layers.{l2.lin_slot}.bias")
        return torch.relu((self._enc1(attr, s1, t1) @ w2.t() + b2) * s2 + t2)

    def classify(self, e):
        l = self.spec.cls_edge[0]
        return e @ self.p(f"classifier.edge_mlp.fc_layers.{l.lin_slot}.weight").t() + \
            self.p(f"classifier.edge_mlp.fc_layers.{l.lin_slot}.bias")

    # ---- full forward ----------------------------------------------------------------------
    def forward(self, x, edge_index, edge_attr, h0=None, n_total=None, e_total=None):
        spec = self.spec
        row, col = edge_index[0], edge_index[1]
        n = x.shape[0] if h0 is None else h0.shape[0]
        e_local = row.numel()
        e_total = e_local if e_total is None else e_total
        H, He = spec.node_dim, spec.edge_dim
        if h0 is None:
            h0 = self.encode_nodes(x, n_total)
        aff = self.edge_encoder_affines(edge_attr, e_total)
        deg = self.reduce(torch.zeros(n, dtype=F64).index_add_(0, row, torch.ones(e_local, dtype=F64)))

        le, ln = spec.upd_edge[0], spec.upd_node[0]
        we = self.p(f"MPNet.edge_model.edge_mlp.fc_layers.{le.lin_slot}.weight")
        be = self.p(f"MPNet.edge_model.edge_mlp.fc_layers.{le.lin_slot}.bias")
        ge = self.p(f"MPNet.edge_model.edge_mlp.fc_layers.{le.bn_slot}.weight")
        bte = self.p(f"MPNet.edge_model.edge_mlp.fc_layers.{le.bn_slot}.bias")
        wn = self.p(f"MPNet.node_model.node_mlp.fc_layers.{ln.lin_slot}.weight")
        bn = self.p(f"MPNet.node_model.node_mlp.fc_layers.{ln.lin_slot}.bias")
        gn = self.p(f"MPNet.node_model.node_mlp.fc_layers.{ln.bn_slot}.weight")
        btn = self.p(f"MPNet.node_model.node_mlp.fc_layers.{ln.bn_slot}.bias")
        hn = (2 if spec.reattach_nodes else 1) * H
        w_pr, w_pc, w_ee = we[:, :hn], we[:, hn:2 * hn], we[:, 2 * hn:]
        w_q, w_a = wn[:, :hn], wn[:, hn:]

        h, e_buf, logits = h0, None, []
        first_cls = spec.num_enc_steps - spec.num_class_steps + 1
        for step in range(1, spec.num_enc_steps + 1):
            hcat = torch.cat([h0, h], 1) if spec.reattach_nodes else h
            pr, pc, q = hcat @ w_pr.t(), hcat @ w_pc.t(), hcat @ w_q.t()          # node_proj kernel
            e_prev = self.e0(edge_attr, aff) if e_buf is None else e_buf
            e_in = torch.cat([self.e0(edge_attr, aff), e_prev], 1) if spec.reattach_edges else e_prev
            # pass A: statistics of z1
            z1 = pr[row] + pc[col] + e_in @ w_ee.t() + be
            st = self.reduce(torch.stack([z1.to(F64).sum(0), (z1.to(F64) ** 2).sum(0)]))
            s1, t1 = bn_affine(st[0], st[1], e_total, ge, bte)
            # pass B: e' and its moments / per-node segment sums
            e_new = torch.relu(z1 * s1 + t1)
            e64 = e_new.to(F64)
            mom = self.reduce(torch.cat([e64.sum(0), (e64.t() @ e64).reshape(-1)]))
            seg = self.reduce(torch.zeros(n, He, dtype=F64).index_add_(0, row, e64))
            # node_stat kernel: statistics of z2 = q[row] + A e' + b by moments
            qb = (q + bn).to(F64)
            a64 = w_a.to(F64)
            proj = seg @ a64.t()                                                    # [N,32] = A . S_i
            sum_z2 = (deg[:, None] * qb + proj).sum(0)
            m2 = mom[He:].reshape(He, He)
            sum_z2sq = (deg[:, None] * qb * qb + 2 * qb * proj).sum(0) + ((a64 @ m2) * a64).sum(1)
            s2, t2 = bn_affine(sum_z2, sum_z2sq, e_total, gn, btn)
            # pass C: messages, aggregation, classifier
            m = torch.relu((q[row] + e_new @ w_a.t() + bn) * s2 + t2)
            idx = row.view(-1, 1).expand_as(m)
            if spec.agg == "max":
                h = torch.zeros(n, H).scatter_reduce(0, idx, m, reduce="amax", include_self=True)
                h = self.reduce_max(h)
            else:
                h = self.reduce(torch.zeros(n, H, dtype=F64).index_add_(0, row, m.to(F64))).float()
                if spec.agg == "mean":
                    h = h / deg.clamp(min=1).float()[:, None]
            e_buf = e_new
            if step >= first_cls:
                logits.append(self.classify(e_new))
        if spec.num_enc_steps == 0:
            logits.append(self.classify(self.e0(edge_attr, aff)))
        return logits, h

    def reduce_max(self, h):
        return h
