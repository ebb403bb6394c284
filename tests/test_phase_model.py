"""The kernel decomposition (tests/phase_model.py: projections, moment-based BatchNorm statistics,
recomputed e0) is algebraically the reference forward: checked against the golden vectors in fp64
distance, and shown to be at least as close to the fp64 truth as the fp32 reference itself."""
import pytest
import torch

from golden_util import Case, case_names
from phase_model import PhaseModel

FWD_CASES = [n for n in case_names()]


@pytest.mark.parametrize("name", FWD_CASES)
def test_phase_model_matches_reference(name):
    c = Case(name)
    m, d = c.model(), c.graph()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    with torch.no_grad():
        logits, h = PhaseModel(sd, m.spec).forward(d.x, d.edge_index, d.edge_attr)
    assert len(logits) == c.meta["n_out"]
    worst = 0.0
    for i, lg in enumerate(logits):
        err64 = (lg[c.sub_idx].double() - c.logits(i, f64=True)).abs().max().item()
        ref_err64 = (c.logits(i).double() - c.logits(i, f64=True)).abs().max().item()
        err_ref = (lg[c.sub_idx] - c.logits(i)).abs().max().item()
        worst = max(worst, err_ref)
        assert err_ref <= 1e-4, f"{name}[{i}] vs fp32 reference: {err_ref:.2e}"
        assert err64 <= max(3 * ref_err64, 2e-5), f"{name}[{i}] vs fp64: {err64:.2e} (reference itself {ref_err64:.2e})"
    assert (h.double() - c.h(f64=True)).abs().max().item() <= 1e-4 * max(1.0, c.h(f64=True).abs().max().item())
