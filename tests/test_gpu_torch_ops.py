"""The registered ops on the GPU: `torch.ops.mtmc_mpn.mp_forward` reproduces the reference fixtures when called
directly (no module), autograd through it equals the module path, and the scatter surface keeps int64 exact."""
import types

import pytest
import torch

from golden_util import ARCH, Case
from mtmc_mpn import engine, torch_ops

pytestmark = pytest.mark.gpu


def _inputs(c):
    d = c.graph()
    ei = d.edge_index.t().contiguous().cuda().t()
    return d, d.x.cuda(), ei, d.edge_attr.cuda()


@pytest.mark.parametrize("name", ["g4_s02_L3", "g2_random_L3_C3", "g5_max", "g5_reattach_nodes"])
def test_mp_forward_op_reproduces_the_fixtures(name):
    c = Case(name)
    m = c.model().cuda().eval()
    _, x, ei, ea = _inputs(c)
    key = torch_ops.config_key(m.model_params, m.arch)
    with torch.no_grad():
        logits, h, tape = torch.ops.mtmc_mpn.mp_forward(x, ei, ea, engine.ordered_params(m), key, False, 0, 0, False)
    assert logits.shape[0] == c.meta["n_out"] and tape.numel() == 0
    for i in range(c.meta["n_out"]):
        assert (logits[i].cpu()[c.sub_idx] - c.logits(i)).abs().max().item() <= 1e-4
    assert (h.cpu().double() - c.h(f64=True)).abs().max().item() <= 1e-4 * max(1.0, c.h(f64=True).abs().max().item())


def test_autograd_through_the_op_equals_the_reference_gradients():
    """G6 (the reference's own gradients, dropout_p = 0): loss over the op's outputs, .backward() through
    torch.library.register_autograd -> torch.ops.mtmc_mpn.mp_backward."""
    from test_gpu_training import loss_of
    c = Case("g6_train_grads")
    m = c.model().cuda().train()
    _, x, ei, ea = _inputs(c)
    params = engine.ordered_params(m)
    key = torch_ops.config_key(m.model_params, m.arch)
    logits, h, tape = torch.ops.mtmc_mpn.mp_forward(x, ei, ea, params, key, True, 123, 0, True)
    assert tape.numel() > 0 and logits.requires_grad
    labels = (torch.rand(c.meta["E"], generator=torch.Generator().manual_seed(c.meta["label_seed"])) < 0.1).long().cuda()
    loss = loss_of({"classified_edges": list(logits.unbind(0))}, labels)
    assert abs(loss.item() - c.meta["loss"]) <= 1e-5 * max(1.0, abs(c.meta["loss"]))
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        g = p.grad.detach().cpu().reshape(-1)
        want = torch.from_numpy(c.blob["grad::" + k])
        got = g[torch.from_numpy(c.blob["grad_idx::" + k])]
        floor = 1e-6 + 2e-4 * max(want.abs().max().item(), 1e-4)
        assert (got - want).abs().max().item() <= floor, k


def test_scatter_ops_and_the_int64_form():
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 40, (3000,), generator=g)
    src_i = torch.randint(0, 2 ** 40, (3000,), generator=g)           # far outside fp32's exact range
    got = torch.ops.mtmc_mpn.scatter_add(src_i.cuda(), idx.cuda(), 0, 41)
    want = torch.zeros(41, dtype=torch.int64).scatter_add_(0, idx, src_i)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    from mtmc_mpn import ops
    ones = torch.ones(3000, dtype=torch.int64)
    assert torch.equal(ops.scatter_add(ones.cuda(), idx.cuda(), dim=0, dim_size=41).cpu(), torch.bincount(idx, minlength=41))
    src_f = torch.randn(3000, 32, generator=g)
    s = torch.ops.mtmc_mpn.scatter_add(src_f.cuda(), idx.cuda(), 0, 41)
    assert (s.cpu() - torch.zeros(41, 32).index_add_(0, idx, src_f)).abs().max().item() <= 1e-4
    v, a = torch.ops.mtmc_mpn.scatter_max(src_f.cuda(), idx.cuda(), 0, 41)
    assert (v[40] == 0).all() and (a[40] == 3000).all()


def test_encode_nodes_op():
    from oracle import mpn_oracle
    c = Case("g1_random_L1")
    m, d = c.model(), c.graph()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    want = mpn_oracle.mlp_forward(d.x, sd, "encoder.node_mlp", mpn_oracle.model_plans(c.params(), ARCH)["enc_node"])
    m = m.cuda()
    key = torch_ops.config_key(m.model_params, m.arch)
    with torch.no_grad():
        got = torch.ops.mtmc_mpn.encode_nodes(d.x.cuda(), engine.ordered_params(m)[:16], key)
    assert (got.cpu() - want).abs().max().item() <= 2e-5


def test_opcheck_schema_and_fake():
    c = Case("g3_cams324_L2")
    m = c.model().cuda().eval()
    _, x, ei, ea = _inputs(c)
    key = torch_ops.config_key(m.model_params, m.arch)
    torch.library.opcheck(torch.ops.mtmc_mpn.mp_forward, (x, ei, ea, engine.ordered_params(m), key, False, 0, 0, False),
                          test_utils=("test_schema", "test_faketensor"))
