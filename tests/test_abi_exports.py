"""CPU test: the built shared library exports every symbol include/mtmc_mpn.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mtmc_mpn import _lib
    header = open(os.path.join(ROOT, "include", "mtmc_mpn.h")).read()
    declared = set(re.findall(r"\b(mtmc_[a-z_0-9]+)\s*\(", header))
    declared -= {"mtmc_scatter_"}                       # prose in a comment
    assert {"mtmc_mpn_forward", "mtmc_mpn_backward", "mtmc_mpn_run_phase", "mtmc_build_graph",
            "mtmc_scatter_add", "mtmc_scatter_mean", "mtmc_scatter_max", "mtmc_mlp_layer_forward",
            "mtmc_postprocess", "mtmc_postprocess_workspace_bytes", "mtmc_cross_entropy_forward",
            "mtmc_cross_entropy_backward", "mtmc_mpn_backward_steps", "mtmc_linear_raw", "mtmc_edge_confusion",
            "mtmc_linear_presplit_raw", "mtmc_mpn_plan_call", "mtmc_cross_entropy_steps_forward", "mtmc_cross_entropy_steps_backward"} <= declared
    assert os.path.exists(_lib.LIB_PATH), "run `python -m mtmc_mpn.build` (or __graft_entry__.build()) first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= declared
    lib.mtmc_mpn_abi_version.restype = ctypes.c_int32
    assert lib.mtmc_mpn_abi_version() == 6


def test_workspace_sizing_needs_no_gpu():
    import mtmc_mpn
    from mtmc_mpn import _lib, engine
    m = mtmc_mpn.MOTMPNet(mtmc_mpn.default_params(num_enc_steps=3), None, "resnet101")
    lib = _lib.load()
    eng = engine.ForwardEngine(m)
    model = _lib.Model()
    # host pointers are fine here: the sizing entry points only read dimensions
    model = eng.model_struct(next(m.parameters()).device)
    small = lib.mtmc_mpn_workspace_bytes(ctypes.byref(model), 450, 150454)
    big = lib.mtmc_mpn_workspace_bytes(ctypes.byref(model), 100000, 10000000)
    train = lib.mtmc_mpn_train_workspace_bytes(ctypes.byref(model), 450, 150454)
    assert 0 < small < train and small < big
    assert 40 * 10000000 < big < 200 * 10000000 + 100000 * 9000
    lay = _lib.WsLayout()
    assert lib.mtmc_mpn_workspace_layout(ctypes.byref(model), 450, 150454, ctypes.byref(lay)) == 0
    assert lay.total_bytes == small and lay.zero_bytes < lay.h0_off


def test_struct_size_guard_refuses_older_callers():
    """ABI v5: a caller compiled against a shorter (older) struct is refused with MTMC_E_ARG instead of being read past
    its end -- checked on the host-only entry points (size queries, plan), which share check_model / check_call_size
    with mtmc_mpn_forward / _run_phase / _backward*."""
    import mtmc_mpn
    from mtmc_mpn import _lib, engine
    m = mtmc_mpn.MOTMPNet(mtmc_mpn.default_params(num_enc_steps=3), None, "resnet101")
    lib = _lib.load()
    eng = engine.ForwardEngine(m)
    model = eng.model_struct(next(m.parameters()).device)
    assert model.struct_bytes == ctypes.sizeof(_lib.Model) and _lib.Call().struct_bytes == ctypes.sizeof(_lib.Call)
    call = _lib.Call()
    call.n_nodes, call.n_edges, call.n_edges_total, call.node_hi = 450, 150454, 150454, 450
    plan = _lib.Plan()
    assert lib.mtmc_mpn_plan_call(ctypes.byref(model), ctypes.byref(call), ctypes.byref(plan)) == 0
    # the ABI-v3 call struct ended at `stream`: 16 bytes shorter
    call.struct_bytes = ctypes.sizeof(_lib.Call) - 16
    assert lib.mtmc_mpn_plan_call(ctypes.byref(model), ctypes.byref(call), ctypes.byref(plan)) == _lib.E_ARG
    assert b"struct_bytes" in lib.mtmc_mpn_last_error()
    call.struct_bytes = 0                                # a caller that never heard of the field
    assert lib.mtmc_mpn_plan_call(ctypes.byref(model), ctypes.byref(call), ctypes.byref(plan)) == _lib.E_ARG
    assert lib.mtmc_mpn_run_phase(ctypes.byref(model), ctypes.byref(call), _lib.PH_BEGIN, 0) == _lib.E_ARG
    assert lib.mtmc_mpn_forward(ctypes.byref(model), ctypes.byref(call)) == _lib.E_ARG      # refused before any launch
    call.struct_bytes = ctypes.sizeof(_lib.Call)
    model.struct_bytes -= 8
    assert lib.mtmc_mpn_workspace_bytes(ctypes.byref(model), 450, 150454) == 0
    assert lib.mtmc_mpn_plan_call(ctypes.byref(model), ctypes.byref(call), ctypes.byref(plan)) == _lib.E_ARG
    assert lib.mtmc_mpn_grad_layout(ctypes.byref(model), None, 0) == 0


def test_struct_sizes_match_the_header(tmp_path):
    """sizeof() of the two call structs as gcc lays them out == the ctypes mirrors in _lib.py."""
    import shutil
    import subprocess
    from mtmc_mpn import _lib
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mtmc_mpn.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", '
                   'sizeof(mtmc_mpn_model), sizeof(mtmc_mpn_call), sizeof(mtmc_ws_layout), sizeof(mtmc_mpn_plan)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(_lib.Model), ctypes.sizeof(_lib.Call), ctypes.sizeof(_lib.WsLayout), ctypes.sizeof(_lib.Plan)]


def test_header_is_plain_c(tmp_path):
    """include/mtmc_mpn.h is a C ABI: it must compile as C99 (and C++11) on its own."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "mtmc_mpn.h"\nint main(void) { return sizeof(mtmc_mpn_call) > 0 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)], check=True)
