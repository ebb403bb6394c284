"""Extracts the fenced ctypes binding of INTEGRATION.md section 2 ("the binding a maintainer would add") and executes it
as it stands, so the documented binding cannot drift from the shipped ABI (round 3 shipped a Call mirror that predated
row_lo / row_hi)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def integration_binding():
    """Namespace of the executed block: lib, Layer, Model, Call, layer(), mpn_forward()."""
    from mtmc_mpn import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("```python\nimport ctypes as C, os, torch\n")
    code = text[start + len("```python\n"):text.index("\n```", start)]
    assert "def mpn_forward(ref_model, data)" in code and "class Call(C.Structure)" in code
    os.environ["MTMC_MPN_LIB"] = _lib.LIB_PATH
    _lib.load()                                     # torch + the ROCm runtime first, as the block's comment asks
    ns = {}
    exec(compile(code, "INTEGRATION.md#binding", "exec"), ns)
    return ns
