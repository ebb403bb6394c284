"""`torch.ops.mtmc_mpn.*` (north_star: "registered as a PyTorch-ROCm custom op"): schemas exist, shape
inference works without a GPU, and there is no CPU kernel behind them (CPU tensors raise instead of computing)."""
import copy

import pytest
import torch

import mtmc_mpn
from mtmc_mpn import engine, torch_ops

ARCH = "resnet101"
OPS = ["mp_forward", "mp_backward", "encode_nodes", "scatter_add", "scatter_mean", "scatter_max"]


def _model(L=3, Cs=2):
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=Cs)
    m = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    return m, torch_ops.config_key(m.model_params, m.arch)


def test_ops_are_registered_with_schemas():
    for name in OPS:
        op = getattr(torch.ops.mtmc_mpn, name)
        assert "mtmc_mpn::" + name in str(op.default._schema)
    assert "Tensor[] params" in str(torch.ops.mtmc_mpn.mp_forward.default._schema)


def test_no_cpu_kernel_behind_the_ops():
    m, key = _model()
    x, ei, ea = torch.randn(8, 2048), torch.randint(0, 8, (2, 20)), torch.rand(20, 2)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mtmc_mpn.mp_forward(x, ei, ea, engine.ordered_params(m), key, False, 0, 0, False)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mtmc_mpn.scatter_add(torch.ones(5, dtype=torch.int64), torch.zeros(5, dtype=torch.int64), 0, 3)
    with pytest.raises(RuntimeError):
        m(__import__("types").SimpleNamespace(x=x, edge_index=ei, edge_attr=ea))


def test_shape_inference_under_fake_tensors():
    from torch._subclasses.fake_tensor import FakeTensorMode
    m, key = _model(L=3, Cs=2)
    params = engine.ordered_params(m)
    assert len(params) == 34
    with FakeTensorMode(allow_non_fake_inputs=True) as mode:
        x, ei, ea = torch.empty(50, 2048), torch.empty(2, 700, dtype=torch.int64), torch.empty(700, 2)
        fp = [mode.from_tensor(p.detach()) for p in params]
        logits, h, tape = torch.ops.mtmc_mpn.mp_forward(x, ei, ea, fp, key, False, 0, 0, False)
        assert logits.shape == (2, 700, 2) and h.shape == (50, 32) and tape.numel() == 0
        flat, dx, dattr = torch.ops.mtmc_mpn.mp_backward(tape, x, ei, ea, fp, key, True, 1, 0, logits, None, True, False)
        assert flat.numel() >= sum(p.numel() for p in params) and dx.shape == x.shape and dattr.numel() == 0
        assert torch.ops.mtmc_mpn.encode_nodes(x, fp[:16], key).shape == (50, 32)
        out = torch.ops.mtmc_mpn.scatter_add(torch.empty(700, dtype=torch.int64), ei[0], 0, 50)
        assert out.shape == (50,) and out.dtype == torch.int64
        v, a = torch.ops.mtmc_mpn.scatter_max(torch.empty(700, 32), ei[0], 0, 50)
        assert v.shape == (50, 32) and a.dtype == torch.int64


def test_fake_tape_has_the_real_tapes_size_and_backward_declares_the_mutation():
    """Under FakeTensorMode / torch.compile the op must describe itself truthfully: with tape=True the third output is
    the training workspace (mtmc_mpn_train_workspace_bytes + 256 bytes, as the real op allocates it), and mp_backward
    writes into it (scratch + one memset), which its schema declares."""
    import ctypes as C
    from torch._subclasses.fake_tensor import FakeTensorMode
    from mtmc_mpn import _lib
    m, key = _model(L=2, Cs=2)
    params = engine.ordered_params(m)
    eng = torch_ops.engine_for(key)
    model = eng.shape_model()
    want = _lib.load().mtmc_mpn_train_workspace_bytes(C.byref(model), 50, 700) + 256
    assert want > 256
    # the placeholder-pointer struct sizes exactly like the one filled from real tensors
    real = engine.ForwardEngine(m).model_struct(torch.device("cpu"))
    assert _lib.load().mtmc_mpn_train_workspace_bytes(C.byref(real), 50, 700) + 256 == want
    with FakeTensorMode(allow_non_fake_inputs=True) as mode:
        x, ei, ea = torch.empty(50, 2048), torch.empty(2, 700, dtype=torch.int64), torch.empty(700, 2)
        fp = [mode.from_tensor(p.detach()) for p in params]
        _, _, tape = torch.ops.mtmc_mpn.mp_forward(x, ei, ea, fp, key, True, 1, 0, True)
        assert tape.dtype == torch.uint8 and tape.numel() == want
    schema = str(torch.ops.mtmc_mpn.mp_backward.default._schema)
    assert "Tensor(a!) tape" in schema


def test_module_rereads_its_parameters_on_every_call():
    """A Parameter object replaced after construction (load_state_dict(assign=True), manual assignment, ...) must be
    the tensor the op receives: the flat list is rebuilt from the module tree per call, never cached."""
    m, _ = _model()
    before = engine.ordered_params(m)
    lin = m.encoder.node_mlp.fc_layers[0]
    lin.weight = torch.nn.Parameter(torch.zeros_like(lin.weight))
    after = engine.ordered_params(m)
    assert after[0] is lin.weight and after[0] is not before[0]
    eng = engine.ForwardEngine(m)
    assert eng.params()[0] is lin.weight
    m.classifier.edge_mlp.fc_layers[0].bias = torch.nn.Parameter(torch.ones(2))
    assert eng.params()[-1] is m.classifier.edge_mlp.fc_layers[0].bias


def test_layout_queries_refuse_a_corrupt_layer_count():
    import ctypes as C
    from mtmc_mpn import _lib
    m, _ = _model()
    model = engine.ForwardEngine(m).model_struct(torch.device("cpu"))
    model.n_enc_layers = 99                                   # would index enc_node[] / the offset array out of bounds
    assert _lib.load().mtmc_mpn_grad_layout(C.byref(model), None, 0) == 0
    call = _lib.Call()
    buf = (C.c_float * 4)()
    rc = _lib.load().mtmc_mpn_backward_flat(C.byref(model), C.byref(call), None, None, buf, 4, None, None)
    assert rc == _lib.E_ARG


def test_python_and_c_agree_on_the_gradient_layout():
    import ctypes as C
    from mtmc_mpn import _lib
    m, key = _model()
    eng = engine.ForwardEngine(m)
    model = eng.model_struct(torch.device("cpu"))           # pointers are never dereferenced by the layout query
    off = (C.c_int64 * 64)()
    total = _lib.load().mtmc_mpn_grad_layout(C.byref(model), off, 64)
    layout, py_total = torch_ops.grad_layout(m.spec)
    assert total == py_total and [o for o, _, _ in layout] == list(off)[:len(layout)]


def test_gradient_layout_covers_every_parameter_once():
    m, key = _model()
    layout, total = torch_ops.grad_layout(m.spec)
    params = engine.ordered_params(m)
    assert [tuple(p.shape) for p in params] == [shp for _, _, shp in layout]
    ends = [o + n for o, n, _ in layout]
    assert all(o % 64 == 0 for o, _, _ in layout) and all(e <= s for e, (s, _, _) in zip(ends, layout[1:])) and ends[-1] <= total
