"""CPU: the struct mirrors of INTEGRATION.md's documented ctypes binding == the header's structs (no compute calls)."""
import ctypes
import os
import shutil
import subprocess

import pytest

from doc_snippet import ROOT, integration_binding


def test_documented_binding_mirrors_the_header(tmp_path):
    from mtmc_mpn import _lib
    ns = integration_binding()
    for name in ("Layer", "Model", "Call"):
        doc, ours = ns[name], getattr(_lib, name)
        assert [f[0] for f in doc._fields_] == [f[0] for f in ours._fields_], name
        assert ctypes.sizeof(doc) == ctypes.sizeof(ours), name
        for f in doc._fields_:
            assert getattr(doc, f[0]).offset == getattr(ours, f[0]).offset, (name, f[0])
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mtmc_mpn.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(mtmc_layer), sizeof(mtmc_mpn_model), sizeof(mtmc_mpn_call), offsetof(mtmc_mpn_call, row_lo), '
                   'offsetof(mtmc_mpn_model, dropout_enc)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(ns["Layer"]), ctypes.sizeof(ns["Model"]), ctypes.sizeof(ns["Call"]),
                   ns["Call"].row_lo.offset, ns["Model"].dropout_enc.offset]
    assert ns["lib"].mtmc_mpn_abi_version() == 6
