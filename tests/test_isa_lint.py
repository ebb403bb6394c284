"""The built library must not contain instruction forms known to miscompute on gfx950 (tools/check_isa.py).

Rule PK-OPSEL: packed-fp32 ops whose low lane selects a source's high dword, inside a kernel with MFMAs, returned 0 in
lanes 48-63 under two waves per SIMD -- the root cause of round 1's "fp16 conversion" failure.  The
compiler's SLP vectoriser creates the form by itself, so the check is on the shipped code objects, not the source."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_isa():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shipped_library_passes_isa_lint():
    from mtmc_mpn import _lib
    ci = _check_isa()
    if not os.path.exists(ci.OBJDUMP):
        import pytest
        pytest.skip("llvm-objdump not available")
    assert os.path.exists(_lib.LIB_PATH), "library not built"
    kernels = list(ci.kernels_of(_lib.LIB_PATH))
    assert len(kernels) > 40 and any("gemm_bn_f16x3" in n for n, _ in kernels)
    assert ci.violations(_lib.LIB_PATH) == []


def test_lint_rule_matches_the_failing_form():
    ci = _check_isa()
    assert ci.PK.search("v_pk_mul_f32 v[42:43], v[8:9], v[38:39] op_sel:[0,1]")
    m = ci.OPSEL_LO.search("v_pk_fma_f32 v[38:39], v[8:9], v[38:39], v[52:53] op_sel:[0,1,0] neg_lo:[0,0,1]")
    assert m and "1" in m.group(1)
    assert ci.OPSEL_LO.search("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]") is None   # measured safe
