"""mtmc_mpn.ops.cross_entropy (fused forward + backward) against torch.nn.functional.cross_entropy."""
import pytest
import torch
import torch.nn.functional as F

from mtmc_mpn import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("classes", [2, 3])
def test_matches_torch(reduction, weighted, classes):
    g = torch.Generator().manual_seed(classes * 10 + weighted)
    n = 20011
    x = (torch.randn(n, classes, generator=g) * 3).to(DEV)
    y = torch.randint(0, classes, (n,), generator=g).to(DEV)
    y[::97] = -100                                                  # ignore_index rows
    w = (torch.rand(classes, generator=g) + 0.5).to(DEV) if weighted else None
    xa, xb = x.clone().requires_grad_(True), x.clone().double().requires_grad_(True)
    got = ops.cross_entropy(xa, y, weight=w, reduction=reduction)
    want = F.cross_entropy(xb, y, weight=w.double() if weighted else None, reduction=reduction)
    scale = max(1.0, float(want.abs().max()))
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    up = torch.randn(want.shape, generator=g, dtype=torch.float64).to(DEV) if reduction == "none" else torch.tensor(1.7, device=DEV, dtype=torch.float64)
    got.backward(up.float())
    want.backward(up)
    gs = max(1e-6, float(xb.grad.abs().max()))
    assert float((xa.grad.double() - xb.grad).abs().max()) <= 2e-6 * gs


def test_training_loss_on_module_outputs():
    """The reference's weighted-CE branch (train.py:118-138) written with the fused op: same value and gradients."""
    g = torch.Generator().manual_seed(3)
    n = 5000
    x = torch.randn(n, 2, generator=g).to(DEV)
    y = (torch.rand(n, generator=g) < 0.02).long().to(DEV)
    n1 = float(y.sum()); n0 = n - n1
    w = torch.tensor([1.0, n0 / n1], device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    per = F.cross_entropy(xb, y, reduction="none")                  # reference formulation, piece by piece
    ref = (per * w[y]).sum() / w[y].sum()
    got = ops.cross_entropy(xa, y, weight=w, reduction="mean")
    assert abs(float(got) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    got.backward(); ref.backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-6 * float(xb.grad.abs().max()) + 1e-9


def test_errors():
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        ops.cross_entropy(torch.zeros(4, 2), torch.zeros(4, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        ops.cross_entropy(torch.zeros(4, 7, device=DEV), torch.zeros(4, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError):
        ops.cross_entropy(torch.zeros(4, 2, device=DEV), torch.zeros(4, dtype=torch.long, device=DEV), reduction="avg")


def test_edge_confusion_counts():
    g = torch.Generator().manual_seed(9)
    n = 50021
    x = torch.randn(n, 2, generator=g)
    x[::50] = 0.25                                                   # ties: argmax picks class 0
    y = (torch.rand(n, generator=g) < 0.1).long()
    pred = torch.argmax(x, 1)
    want = [int(((pred == 1) & (y == 1)).sum()), int(((pred == 1) & (y == 0)).sum()),
            int(((pred == 0) & (y == 0)).sum()), int(((pred == 0) & (y == 1)).sum())]
    got = ops.edge_confusion(x.to(DEV), y.to(DEV)).cpu().tolist()
    assert got == want
    # the reference's FPR (train.py:100-102)
    fp = torch.sum(pred[y == 0]); tn = pred[y == 0].shape[0] - fp
    assert abs(got[1] / (got[1] + got[2]) - float(fp / (fp + tn))) < 1e-6      # torch divides in float32


@pytest.mark.parametrize("reduction,weighted", [("mean", True), ("mean", False), ("sum", True)])
def test_cross_entropy_steps_equals_the_sum_over_steps(reduction, weighted):
    """The training loop's loss over the classified steps in one pass: value and gradients equal torch's per-step sum;
    a list that is not one logits block falls back to the per-step path."""
    import mtmc_mpn
    g = torch.Generator(device="cuda").manual_seed(11)
    S, E, C = 3, 5003, 2
    block = torch.randn(S, E, C, device="cuda", generator=g).requires_grad_(True)
    labels = (torch.rand(E, device="cuda", generator=g) < 0.2).long()
    w = torch.tensor([1.0, 4.2], device="cuda") if weighted else None
    steps = [block[i] for i in range(S)]
    loss = mtmc_mpn.cross_entropy_steps(steps, labels, weight=w, reduction=reduction)
    loss.backward()
    ref_in = block.detach().clone().double().requires_grad_(True)
    ref = sum(torch.nn.functional.cross_entropy(ref_in[i], labels, weight=None if w is None else w.double(), reduction=reduction)
              for i in range(S))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert (block.grad.double() - ref_in.grad).abs().max().item() <= 2e-6 * max(1.0, ref_in.grad.abs().max().item())
    # separate tensors: same value through the fallback
    sep = [block.detach()[i].clone() for i in range(S)]
    loss2 = mtmc_mpn.cross_entropy_steps(sep, labels, weight=w, reduction=reduction)
    assert abs(loss2.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))


def test_cross_entropy_steps_on_the_outputs_of_a_training_forward():
    import copy
    import types
    import mtmc_mpn
    from mtmc_mpn import graphs
    d = graphs.camera_graph((12, 9, 10), seed=3)
    torch.manual_seed(0)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(mtmc_mpn.default_params(num_enc_steps=3, num_class_steps=3)), None, "resnet101").cuda().train()
    data = types.SimpleNamespace(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda())
    labels = (torch.rand(d.edge_index.shape[1], generator=torch.Generator().manual_seed(1)) < 0.2).long().cuda()
    grads = []
    for fused in (True, False):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(7)                                   # same Dropout masks
        out, _ = m(data)
        steps = out["classified_edges"]
        loss = (mtmc_mpn.cross_entropy_steps(steps, labels) if fused
                else sum(mtmc_mpn.cross_entropy(s, labels) for s in steps))
        loss.backward()
        grads.append((loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * max(1.0, abs(grads[1][0]))
    for k in grads[0][1]:
        a, b = grads[0][1][k], grads[1][1][k]
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-6, k   # (atomics order: last-bit noise)
