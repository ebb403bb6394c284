"""HIP post-processing (mtmc_postprocess) against the oracle / the reference-generated fixtures: exact."""
import pytest
import torch

import mtmc_mpn
import pp_cases
from oracle import postprocess_oracle as po
from pp_util import NAMES, PpCase

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", NAMES)
def test_fixture_given_probabilities(name):
    """The reference run's own probabilities in -> the reference's predictions and cluster numbering out."""
    c = PpCase(name)
    out = mtmc_mpn.postprocess(None, c.s.edge_index.to(DEV), c.s.n_nodes, c.s.n_cams, *c.flags,
                               preds_prob=c.prob1.to(DEV), predictions=c.pred_in.to(DEV))
    assert torch.equal(out.predictions.cpu(), c.pred_out)
    assert torch.equal(out.ID_pred.cpu(), c.ids)
    assert out.info["active_in"] == c.meta["active_in"] and out.info["active_out"] == c.meta["active_out"]
    assert out.info["clusters"] == c.meta["clusters_out"]


@pytest.mark.parametrize("name", ["pp2_noisy", "pp4_perm", "pp6_six_cams"])
def test_fused_softmax_argmax(name):
    """From logits: p1 within 1e-6 of torch.softmax, argmax identical; the graph logic exact on the device's p1."""
    c = PpCase(name)
    out = mtmc_mpn.postprocess(c.s.logits.to(DEV), c.s.edge_index.to(DEV), c.s.n_nodes, c.s.n_cams, *c.flags)
    p1 = out.preds_prob1.cpu()
    ref_prob, ref_pred = po.classify(c.s.logits)
    assert torch.max(torch.abs(p1 - ref_prob[:, 1])) <= 1e-6
    ids, pred = po.post_processing(c.s.n_cams, ref_pred, c.s.edge_index, c.s.n_nodes,
                                   torch.stack([1 - p1, p1], dim=1), *c.flags)
    assert torch.equal(out.predictions.cpu(), pred)
    assert torch.equal(out.ID_pred.cpu(), ids)


@pytest.mark.parametrize("seed", range(8))
def test_random_scenarios_vs_oracle(seed):
    g = torch.Generator().manual_seed(100 + seed)
    kw = dict(n_ids=int(torch.randint(5, 40, (1,), generator=g)), n_cams=int(torch.randint(2, 7, (1,), generator=g)),
              seed=200 + seed, fp_rate=float(torch.rand(1, generator=g)) * 0.03,
              fn_rate=float(torch.rand(1, generator=g)) * 0.2, pair_fp=float(torch.rand(1, generator=g)) * 0.03,
              quant=[0.0, 0.5, 2.0][seed % 3], perm=bool(seed & 1))
    s = pp_cases.scenario(**kw)
    flags = (bool(seed & 1) or seed > 3, bool(seed & 2) or seed > 3, bool(seed & 4) or seed > 5)
    out = mtmc_mpn.postprocess(s.logits.to(DEV), s.edge_index.to(DEV), s.n_nodes, s.n_cams, *flags)
    p1 = out.preds_prob1.cpu()
    ids, pred = po.post_processing(s.n_cams, torch.argmax(s.logits, 1), s.edge_index, s.n_nodes,
                                   torch.stack([1 - p1, p1], dim=1), *flags)
    assert torch.equal(out.predictions.cpu(), pred), kw
    assert torch.equal(out.ID_pred.cpu(), ids), kw


def test_no_active_edges_and_tiny():
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]], device=DEV)
    logits = torch.tensor([[2.0, -1.0]] * 4, device=DEV)
    out = mtmc_mpn.postprocess(logits, ei, 3, 2)
    assert out.predictions.sum().item() == 0 and out.ID_pred.cpu().tolist() == [0, 1, 2]
    logits = torch.tensor([[-1.0, 2.0], [-1.0, 2.0], [2.0, -1.0], [-1.0, 2.0]], device=DEV)   # 0<->1 kept, 2->1 cut
    out = mtmc_mpn.postprocess(logits, ei, 3, 2)
    assert out.predictions.cpu().tolist() == [1, 1, 0, 0] and out.ID_pred.cpu().tolist() == [0, 0, 1]


def test_transposed_edge_index_view_and_global_memory_path():
    """[E,2].T views (element stride 2) and graphs too large for the LDS-resident state take the same code."""
    s = pp_cases.scenario(n_ids=700, n_cams=4, seed=31, fp_rate=0.0002, fn_rate=0.05, pair_fp=0.0002)
    assert s.n_nodes > 2048
    pairs = s.edge_index.t().contiguous().to(DEV)
    out = mtmc_mpn.postprocess(s.logits.to(DEV), pairs.t(), s.n_nodes, s.n_cams)
    p1 = out.preds_prob1.cpu()
    ids, pred = po.post_processing(s.n_cams, torch.argmax(s.logits, 1), s.edge_index, s.n_nodes,
                                   torch.stack([1 - p1, p1], dim=1))
    assert torch.equal(out.predictions.cpu(), pred)
    assert torch.equal(out.ID_pred.cpu(), ids)


def test_capacity_overflow_is_reported():
    s = pp_cases.scenario(n_ids=20, n_cams=4, seed=5, fp_rate=0.2)
    with pytest.raises(RuntimeError, match="max_active"):
        mtmc_mpn.postprocess(s.logits.to(DEV), s.edge_index.to(DEV), s.n_nodes, s.n_cams, max_active=8)


def test_out_of_range_node_ids_are_clamped_and_reported():
    ei = torch.tensor([[0, 1, 1, 7], [1, 0, 2, 1]], device=DEV)              # node 7 does not exist (N = 3)
    logits = torch.tensor([[-1.0, 2.0]] * 4, device=DEV)
    with pytest.raises(RuntimeError, match="outside"):
        mtmc_mpn.postprocess(logits, ei, 3, 2)
    out = mtmc_mpn.postprocess(logits, ei, 3, 2, check=False)                # no fault, no exception without the check
    torch.cuda.synchronize()
    assert out.info_dev.cpu()[3].item() == 4
