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


def test_encode_nodes_refuses_parameters_it_cannot_hand_to_the_kernels():
    """Every pointer of `params` reaches the GEMM kernels as is: CPU, half-precision, strided or wrongly shaped tensors
    must raise BEFORE any library call (they would be wild reads on the device)."""
    c = Case("g1_random_L1")
    m, d = c.model().cuda(), c.graph()
    key = torch_ops.config_key(m.model_params, m.arch)
    x = d.x.cuda()
    good = engine.ordered_params(m)[:16]

    def with_(i, t):
        p = list(good)
        p[i] = t
        return p
    bad = [with_(0, good[0].detach().cpu()),                       # host pointer
           with_(1, good[1].detach().half()),                      # wrong dtype
           with_(0, good[0].detach().t().contiguous().t()),        # right shape, strided
           with_(4, good[4].detach()[:, :-32].contiguous()),       # layer 1 weight [512, 992]
           with_(2, good[2].detach()[:-1].contiguous())]           # gamma one short
    for p in bad:
        with pytest.raises(RuntimeError):
            torch.ops.mtmc_mpn.encode_nodes(x, p, key)
    with pytest.raises(RuntimeError):
        torch.ops.mtmc_mpn.encode_nodes(x.double(), good, key)
    torch.cuda.synchronize()
    assert torch.isfinite(torch.ops.mtmc_mpn.encode_nodes(x, good, key)).all()      # the device is still healthy


def test_scatter_ops_refuse_an_index_on_another_device():
    src = torch.randn(100, 4, device="cuda")
    idx_cpu = torch.randint(0, 10, (100,))
    for op in (torch.ops.mtmc_mpn.scatter_add, torch.ops.mtmc_mpn.scatter_mean, torch.ops.mtmc_mpn.scatter_max):
        with pytest.raises(RuntimeError):
            op(src, idx_cpu, 0, 10)
    with pytest.raises(RuntimeError):
        torch.ops.mtmc_mpn.scatter_add(src, torch.rand(100, device="cuda"), 0, 10)    # float "index"
    torch.cuda.synchronize()


def test_a_replaced_parameter_is_the_one_used_and_the_one_that_gets_the_gradient():
    """The module re-reads its parameters on every call: after a forward, swap in a new Parameter object (what
    load_state_dict(assign=True) or `layer.weight = nn.Parameter(...)` do) -- the next forward must compute with it and
    backward must deliver the gradient to it, not to the tensor the first call saw."""
    c = Case("g3_cams324_L2")
    m = c.model().cuda().eval()
    _, x, ei, ea = _inputs(c)
    data = types.SimpleNamespace(x=x, edge_index=ei, edge_attr=ea)
    with torch.no_grad():
        first = m(data)[0]["classified_edges"][-1].clone()
    cls = m.classifier.edge_mlp.fc_layers[0]
    old = cls.bias
    cls.bias = torch.nn.Parameter(old.detach() + 1.0)
    with torch.no_grad():
        second = m(data)[0]["classified_edges"][-1]
    assert (second - first - 1.0).abs().max().item() <= 1e-5         # the classifier bias shifts every logit by exactly 1
    sd = {k: v.detach().clone() + 0.25 for k, v in m.state_dict().items() if k.endswith("fc_layers.0.bias") and "classifier" in k}
    m.load_state_dict(sd, strict=False, assign=True)
    with torch.no_grad():
        third = m(data)[0]["classified_edges"][-1]
    assert (third - second - 0.25).abs().max().item() <= 1e-5
    out = m(data)[0]["classified_edges"][-1]
    out.sum().backward()
    assert cls.bias.grad is not None and old.grad is None
    assert torch.allclose(cls.bias.grad, torch.full_like(cls.bias, float(out.shape[0])))
