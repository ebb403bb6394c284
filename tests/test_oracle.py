"""CPU tests: the oracle against the golden vectors generated from the reference itself,
the third-party scatter semantics, and the drop-in module's parameter contract."""
import copy

import pytest
import torch

import mtmc_mpn
from golden_util import ARCH, Case, case_names, sha
from oracle import mpn_oracle

CASES = case_names()


def test_fixtures_present():
    assert len(CASES) >= 15


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    """oracle fp32 == reference fp32 (bit for bit on the generating CPU type, else 2e-6),
    oracle fp64 == reference fp64 to 1e-12."""
    c = Case(name)
    m, d = c.model(), c.graph()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    same_inputs = c.inputs_match_reference_run(d)
    out, h = mpn_oracle.forward(sd, c.params(), ARCH, d.x, d.edge_index, d.edge_attr)
    out64, h64 = mpn_oracle.forward(sd, c.params(), ARCH, d.x, d.edge_index, d.edge_attr, dtype=torch.float64)
    assert len(out["classified_edges"]) == c.meta["n_out"]
    bit_exact = same_inputs and torch.backends.cpu.get_cpu_capability() == c.meta["cpu_capability"] \
        and torch.__version__ == c.meta["torch"]
    for i, (a, a64) in enumerate(zip(out["classified_edges"], out64["classified_edges"])):
        assert a.shape == (c.meta["E"], 2) and a.dtype == torch.float32
        if bit_exact:
            assert sha(a) == c.meta["logits_sha"][i], f"{name}: oracle no longer bit-equal to the reference"
        assert torch.allclose(a[c.sub_idx], c.logits(i), rtol=0, atol=2e-6 if not bit_exact else 0)
        assert torch.allclose(a64[c.sub_idx], c.logits(i, f64=True), rtol=0, atol=1e-9 if not bit_exact else 1e-12)
        assert abs(float(a.double().sum()) - c.meta["logits_sum"][i]) <= 1e-6 * max(1.0, c.meta["logits_abssum"][i])
    if bit_exact:
        assert sha(h) == c.meta["h_sha"]
    assert torch.allclose(h, c.h(), rtol=0, atol=0 if bit_exact else 2e-5)
    assert torch.allclose(h64, c.h(f64=True), rtol=1e-10, atol=1e-9)


def test_survey_anchor_values():
    """Sanity anchors recorded in SURVEY.md 8(c) for config 1 (weights seed 0, inputs seed 1)."""
    c = Case("g1_random_L1")
    assert abs(float(c.logits(0)[0, 0]) - 0.14883381) < 1e-6 and abs(float(c.logits(0)[0, 1]) + 0.41112748) < 1e-6
    assert abs(c.meta["logits_sum"][0] + 268.903075) < 1e-4 and c.meta["n_pos"][0] == 30
    c3 = Case("g2_random_L3_C1")
    assert abs(float(c3.logits(0)[0, 0]) - 0.18001047) < 1e-6 and c3.meta["n_pos"][0] == 8


def test_scatter_known_answers():
    """Hand-computed 5-edge case with a repeated row (1), an empty row (2) and negative values:
    pins the torch_scatter 2.0.8 semantics the oracle restates (sum / mean / max-with-0-fill)."""
    m = torch.tensor([[1., -2.], [3., 4.], [-5., -6.], [7., 8.], [0.5, -0.5]])
    row = torch.tensor([1, 1, 3, 0, 1])
    s = mpn_oracle.aggregate(m, row, 4, "sum")
    assert torch.equal(s, torch.tensor([[7., 8.], [4.5, 1.5], [0., 0.], [-5., -6.]]))
    mean = mpn_oracle.aggregate(m, row, 4, "mean")
    assert torch.allclose(mean, torch.tensor([[7., 8.], [1.5, 0.5], [0., 0.], [-5., -6.]]))
    mx = mpn_oracle.aggregate(m, row, 4, "max")
    assert torch.equal(mx, torch.tensor([[7., 8.], [3., 4.], [0., 0.], [-5., -6.]]))
    with pytest.raises(AssertionError):
        mpn_oracle.aggregate(m, row, 4, "median")


def test_state_dict_contract():
    """34 tensors, 2 697 750 parameters, no buffers, reference key names/shapes."""
    m = mtmc_mpn.MOTMPNet(mtmc_mpn.default_params(), None, ARCH)
    sd = m.state_dict()
    assert len(sd) == 34 and sum(v.numel() for v in sd.values()) == 2697750
    assert len(list(m.buffers())) == 0
    want = {"encoder.node_mlp.fc_layers.0.weight": (1024, 2048), "encoder.node_mlp.fc_layers.13.bias": (32,),
            "encoder.edge_mlp.fc_layers.0.weight": (4, 2), "encoder.edge_mlp.fc_layers.4.weight": (4, 4),
            "classifier.edge_mlp.fc_layers.0.weight": (2, 4),
            "MPNet.edge_model.edge_mlp.fc_layers.0.weight": (4, 68), "MPNet.edge_model.edge_mlp.fc_layers.1.bias": (4,),
            "MPNet.node_model.node_mlp.fc_layers.0.weight": (32, 36), "MPNet.node_model.node_mlp.fc_layers.1.weight": (32,)}
    for k, shp in want.items():
        assert tuple(sd[k].shape) == shp, k
    # strict round trip through a reference-style checkpoint dict, including a 'module.' prefix strip
    ck = {"module." + k: v.clone() for k, v in sd.items()}
    m2 = mtmc_mpn.MOTMPNet(mtmc_mpn.default_params(), None, ARCH)
    m2.load_state_dict({k[len("module."):]: v for k, v in ck.items()}, strict=True)
    re = mtmc_mpn.MOTMPNet(mtmc_mpn.default_params(reattach_initial_nodes=True, reattach_initial_edges=True), None, ARCH)
    assert tuple(re.state_dict()["MPNet.edge_model.edge_mlp.fc_layers.0.weight"].shape) == (4, 136)
    assert tuple(re.state_dict()["MPNet.node_model.node_mlp.fc_layers.0.weight"].shape) == (32, 68)


def test_constructor_errors_and_config_side_effect():
    p = mtmc_mpn.default_params(node_agg_fn="median")
    with pytest.raises(AssertionError):
        mtmc_mpn.MOTMPNet(p, None, ARCH)
    p = mtmc_mpn.default_params()
    mtmc_mpn.MOTMPNet(p, None, ARCH)
    # the reference merges the node-encoder dict into the edge-encoder dict in place (models/mpn.py:169)
    assert p["encoder_feats_dict"]["edges"]["node_in_dim"] == 2048
    p2 = copy.deepcopy(mtmc_mpn.DEFAULT_GRAPH_NET_PARAMS)
    p2["node_model_feats_dict"]["fc_dims"] = [56, 32]
    with pytest.raises(NotImplementedError):
        mtmc_mpn.MOTMPNet(p2, None, ARCH)


def test_forward_refuses_cpu_tensors():
    """No CPU fallback: the product path must fail loudly off-GPU."""
    c = Case("g3_cams324_L2")
    m, d = c.model(), c.graph()
    with pytest.raises(RuntimeError):
        m(d)
