"""The weight-plane cache verifies itself (csrc/split_body.h, engine.weight_plane_cache): every way of changing a weight that
fooled round 4's host-side (data_ptr, _version) key -- ADVICE round 4 -- must give the NEW weights' logits, on the few-row
path (every layer reads cached planes) and on the many-row path (layers 0 and 1 do)."""
import copy
import gc
import types

import pytest
import torch

import mtmc_mpn
from mtmc_mpn import _lib, engine, graphs, torch_ops

pytestmark = pytest.mark.gpu
ARCH = "resnet101"
TOL = 1e-4


def _oracle(model, params, d):
    from oracle import mpn_oracle
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        out, h = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, d.x.cpu(), d.edge_index.cpu(), d.edge_attr.cpu(), dtype=torch.float64)
    return out["classified_edges"][-1], h


def _fwd(model, d):
    with torch.no_grad():
        out, h = model(d)
    torch.cuda.synchronize()
    return out["classified_edges"][-1].cpu(), h.cpu()


def _graph(cams=(40, 30, 35), seed=3):
    d = graphs.camera_graph(cams, seed=seed)
    return types.SimpleNamespace(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda())


def _model(seed, L=2):
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=1)
    torch.manual_seed(seed)
    return mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval().cuda(), params


def _assert_matches(model, params, d, tag):
    got, h = _fwd(model, d)
    want, h64 = _oracle(model, params, d)
    err = (got.double() - want).abs().max().item()
    assert err <= TOL, f"{tag}: max |dlogit| {err:.3e}"
    assert (h.double() - h64).abs().max().item() <= 1e-4 * max(1.0, h64.abs().max().item()), tag


def test_few_row_path_is_taken_and_uses_the_cache():
    m, params = _model(0)
    d = _graph()
    plan = engine.ForwardEngine(m).plan(d.x.shape[0], d.edge_index.shape[1])
    assert plan.enc_kernel[0] == _lib.GEMM_FEW_L0 and plan.enc_kernel[1] == _lib.GEMM_FEW_WAVE
    _assert_matches(m, params, d, "first forward (every chunk derived)")
    _assert_matches(m, params, d, "second forward (every chunk verified, none derived)")


@pytest.mark.parametrize("layer", [0, 1, 2, 3])
def test_write_through_param_data_is_seen(layer):
    """`param.data` writes do not bump Parameter._version: round 4's key called the planes valid and multiplied by the OLD
    weights without any error."""
    m, params = _model(1)
    d = _graph()
    _assert_matches(m, params, d, "before")
    w = engine.ordered_params(m)[4 * layer]
    v0 = w._version
    w.data[5, 7] += 0.5                                         # ONE element of one chunk
    w.data[w.shape[0] - 1].mul_(-1.0)                           # ... and a whole row of the last chunk
    assert w._version == v0
    _assert_matches(m, params, d, f"after a param.data write to encoder layer {layer}")


def test_new_module_at_recycled_addresses():
    """`del model; model = load(ckpt2)`: the caching allocator hands back the same addresses at the same _version; the engine
    (and its cache buffer) is process-global per configuration."""
    d = _graph()
    m1, params = _model(2)
    _assert_matches(m1, params, d, "model 1")
    ptrs1 = [p.data_ptr() for p in engine.ordered_params(m1)[:4]]
    del m1
    gc.collect()
    m2, params = _model(3)
    ptrs2 = [p.data_ptr() for p in engine.ordered_params(m2)[:4]]
    _assert_matches(m2, params, d, "model 2" + (" at recycled addresses" if ptrs1 == ptrs2 else ""))


def test_failed_call_then_good_call():
    """A call that fails before anything is launched must not leave a cache that a later call trusts."""
    m, params = _model(4)
    d = _graph()
    bad = types.SimpleNamespace(x=d.x[:1].clone(), edge_index=d.edge_index[:, :0], edge_attr=d.edge_attr[:0])
    with pytest.raises((ValueError, RuntimeError)):
        m(bad)                                                  # BatchNorm over one row: MTMC_E_ROWS
    _assert_matches(m, params, d, "good call after a failed one")
    w = engine.ordered_params(m)[0]
    w.data.mul_(0.5)
    with pytest.raises((ValueError, RuntimeError)):
        m(bad)
    _assert_matches(m, params, d, "good call after a failed one and a weight change")


def test_optimizer_step_and_load_state_dict():
    m, params = _model(5)
    d = _graph()
    _assert_matches(m, params, d, "before")
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01 * torch.randn_like(p))
    _assert_matches(m, params, d, "after in-place updates of all 34 parameters")
    other, _ = _model(6)
    m.load_state_dict(other.state_dict())
    _assert_matches(m, params, d, "after load_state_dict")


def test_without_the_cache_the_split_k_kernels_still_run():
    m, params = _model(7)
    d = _graph()
    m.cache_weight_planes = False
    _assert_matches(m, params, d, "cache_weight_planes = False")
    got_off, _ = _fwd(m, d)
    m.cache_weight_planes = True
    got_on, _ = _fwd(m, d)
    assert (got_on - got_off).abs().max().item() <= 2e-5       # two kernel families, one answer


def test_captured_graph_follows_param_data_writes():
    """Under graph capture round 4 never used the cache; now the verification itself is part of the captured forward."""
    m, params = _model(8)
    d = _graph()
    replay = m.capture(d)
    out, _ = replay()
    torch.cuda.synchronize()
    want, _ = _oracle(m, params, d)
    assert (out["classified_edges"][-1].cpu().double() - want).abs().max().item() <= TOL
    engine.ordered_params(m)[4].data[3].add_(0.25)
    out, _ = replay()
    torch.cuda.synchronize()
    want2, _ = _oracle(m, params, d)
    assert (want2 - want).abs().max().item() > 1e-3             # (the change matters)
    assert (out["classified_edges"][-1].cpu().double() - want2).abs().max().item() <= TOL


def test_many_row_graph_sees_weight_changes():
    """8200 node rows: layer 0 on pre-split weights and layer 1 on the role-split kernel read the cached planes."""
    m, params = _model(9, L=1)
    g = graphs.stress_graph(8200, 30_000, seed=5)
    d = types.SimpleNamespace(x=g.x.cuda(), edge_index=g.edge_index.cuda(), edge_attr=g.edge_attr.cuda())
    plan = engine.ForwardEngine(m).plan(8200, d.edge_index.shape[1])
    assert plan.enc_kernel[:2] == [_lib.GEMM_PRESPLIT_256, _lib.GEMM_STAGED_128]
    _assert_matches(m, params, d, "before")
    for layer in (0, 1):
        engine.ordered_params(m)[4 * layer].data[11].mul_(1.5)
    _assert_matches(m, params, d, "after param.data writes to layers 0 and 1")
