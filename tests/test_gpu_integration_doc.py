"""The binding INTEGRATION.md section 2 documents, executed as it stands against MOTMPNet.forward (reference call
site inference.py:469: `outputs, latent_node_feats = mpn_model(data)`)."""
import copy
import types

import pytest
import torch

import mtmc_mpn
from mtmc_mpn import graphs

from doc_snippet import integration_binding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,Cs,transposed_view", [(3, 1, True), (2, 2, False)])
def test_documented_binding_runs_and_matches_the_module(L, Cs, transposed_view):
    ns = integration_binding()
    params = mtmc_mpn.default_params(num_enc_steps=L, num_class_steps=Cs)
    torch.manual_seed(11)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval().cuda()
    d = graphs.camera_graph((21, 17, 19, 12), seed=9)
    ei = d.edge_index.cuda()
    if transposed_view:                               # the callers' [E,2].T view (inference.py:446-450)
        ei = ei.t().contiguous().t()
    data = types.SimpleNamespace(x=d.x.cuda(), edge_index=ei, edge_attr=d.edge_attr.cuda())
    with torch.no_grad():
        got, got_h = ns["mpn_forward"](model, data)
        want, want_h = model(data)
    torch.cuda.synchronize()
    assert len(got["classified_edges"]) == len(want["classified_edges"]) == min(L, Cs)
    for a, b in zip(got["classified_edges"], want["classified_edges"]):
        assert (a - b).abs().max().item() <= 2e-6
    assert (got_h - want_h).abs().max().item() <= 1e-5 * max(1.0, want_h.abs().max().item())


def test_short_call_struct_is_refused_not_read_past():
    """What round 3's documented binding did: a Call that ends at `stream`.  ABI v5 refuses it."""
    import ctypes as C
    ns = integration_binding()
    lib, Model, Call = ns["lib"], ns["Model"], ns["Call"]
    short_fields = [f for f in Call._fields_ if f[0] not in ("row_lo", "row_hi")]
    Short = type("Short", (C.Structure,), {"_fields_": short_fields})
    params = mtmc_mpn.default_params(num_enc_steps=1, num_class_steps=1)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval().cuda()
    from mtmc_mpn import engine
    m = engine.ForwardEngine(model).model_struct(torch.device("cuda", torch.cuda.current_device()))
    m2 = Model.from_buffer_copy(m)
    c = Short(struct_bytes=C.sizeof(Short), n_nodes=10, n_edges=20, n_edges_total=20, node_hi=10)
    lib.mtmc_mpn_forward.argtypes = [C.POINTER(Model), C.c_void_p]
    assert lib.mtmc_mpn_forward(C.byref(m2), C.cast(C.pointer(c), C.c_void_p)) == -1
    assert b"struct_bytes" in lib.mtmc_mpn_last_error()
    lib.mtmc_mpn_forward.argtypes = [C.POINTER(Model), C.POINTER(Call)]
