"""GPU tests of the graph builder (SURVEY.md 8(f)-1,2) against the CPU oracle that restates the reference's
inference.py:402-456 with the same torch calls: integer outputs exact, floating outputs to fp32 rounding."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import mtmc_mpn
from golden_util import ARCH, GOLDEN_DIR
from mtmc_mpn import graphs

pytestmark = pytest.mark.gpu


def check(feats, cams, labels, l2norm=True):
    from oracle import graph_oracle
    x_ref, ei_ref, attr_ref, lab_ref = graph_oracle.build(feats, cams, labels, l2norm)
    g = mtmc_mpn.build_graph(feats.cuda(), cams, labels, l2norm)
    assert g.edge_index.shape == ei_ref.shape and g.edge_index.dtype == torch.int64
    assert not g.edge_index.is_contiguous()                       # the callers' [E,2].T view, as in inference.py:413
    assert torch.equal(g.edge_index.cpu(), ei_ref)                # bit-exact integer work
    if labels is not None:
        assert torch.equal(g.edge_labels.cpu(), lab_ref)
    assert (g.x.cpu() - x_ref).abs().max().item() <= 1e-6 * max(1.0, x_ref.abs().max().item())
    err = (g.edge_attr.cpu() - attr_ref).abs().max(0).values
    # distance from the Gram form: fp32 rounding of a K=2048 dot product times the cancellation (|a|^2+|b|^2)/d^2
    assert err[0].item() <= 2e-5 * max(1.0, attr_ref[:, 0].abs().max().item()), f"distance err {err[0]:.2e}"
    assert err[1].item() <= 5e-6, f"cosine err {err[1]:.2e}"
    return g, (x_ref, ei_ref, attr_ref)


def test_s02_topology_inference_order():
    cams = np.repeat(np.arange(4), graphs.S02_GT_CAMS)
    feats = torch.randn(cams.size, 2048, generator=torch.Generator().manual_seed(2))
    labels = torch.randint(0, 145, (cams.size,), generator=torch.Generator().manual_seed(3)).numpy()
    g, _ = check(feats, cams, labels)
    assert g.edge_index.shape[1] == 150454


def test_training_order_and_near_duplicates():
    """Nodes ordered by identity then camera (train.py:295-302): cameras interleave; same-identity tracklets get
    nearly identical features, the case where the Gram form of the distance cancels the most."""
    with open(os.path.join(GOLDEN_DIR, "train_tracklets.json")) as f:
        tr = json.load(f)["tracklets"]
    ids = sorted({t[1] for t in tr})[:60]
    nodes = [(c, i) for i in ids for (c, j) in sorted(tr) if j == i]
    cams = np.array([c for c, _ in nodes])
    labels = np.array([i for _, i in nodes])
    gen = torch.Generator().manual_seed(5)
    base = {i: torch.randn(2048, generator=gen) for i in ids}
    feats = torch.stack([base[i] + 0.05 * torch.randn(2048, generator=gen) for _, i in nodes])
    check(feats, cams, labels)
    check(feats.abs(), cams, None, l2norm=False)                  # un-normalised branch, no labels


def test_two_nodes_two_cameras_and_single_camera():
    feats = torch.randn(2, 2048, generator=torch.Generator().manual_seed(1))
    check(feats, np.array([3, 7]), np.array([1, 1]))
    g = mtmc_mpn.build_graph(torch.randn(5, 2048).cuda(), np.zeros(5, dtype=int))   # one camera: no edges
    assert g.edge_index.shape == (2, 0) and g.edge_attr.shape == (0, 2)
    with pytest.raises(RuntimeError):
        mtmc_mpn.build_graph(torch.randn(5, 2048), np.zeros(5, dtype=int))          # CPU tensor refused


def test_built_graph_feeds_the_mpn_like_the_reference_pipeline():
    from oracle import graph_oracle, mpn_oracle
    cams = np.repeat(np.arange(3), (20, 17, 25))
    feats = torch.randn(cams.size, 2048, generator=torch.Generator().manual_seed(8))
    params = mtmc_mpn.default_params(num_enc_steps=2, num_class_steps=1)
    torch.manual_seed(0)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, ei, attr, _ = graph_oracle.build(feats, cams)
    with torch.no_grad():
        want, _ = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, x, ei, attr)
        got, _ = m.cuda()(mtmc_mpn.build_graph(feats.cuda(), cams))
    assert (got["classified_edges"][0].cpu() - want["classified_edges"][0]).abs().max().item() <= 1e-4
