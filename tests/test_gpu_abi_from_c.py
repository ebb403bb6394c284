"""The C ABI without PyTorch: tests/abi/forward_from_c.cpp (hipMalloc + mtmc_mpn_forward) against the Python module."""
import copy
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

import mtmc_mpn
from mtmc_mpn import _lib, engine, graphs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_from_a_plain_hip_program(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "forward_from_c")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "forward_from_c.cpp"),
                    "-L", libdir, "-lmtmc_mpn", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    params = mtmc_mpn.default_params(num_enc_steps=2, num_class_steps=2)
    torch.manual_seed(4)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
    d = graphs.camera_graph((17, 12, 15), seed=6)
    n, e = d.x.shape[0], d.edge_index.shape[1]
    tensors = []
    for _, lin, bn, _ in engine.ForwardEngine(model).param_layers():
        tensors += [lin.weight, lin.bias] + ([bn.weight, bn.bias] if bn is not None else [])
    blob = str(tmp_path / "in.bin")
    with open(blob, "wb") as f:
        f.write(struct.pack("<6q", n, e, 2, 2, 2048, len(tensors)))
        for t in tensors:
            a = t.detach().numpy().astype("<f4").ravel()
            f.write(struct.pack("<q", a.size))
            f.write(a.tobytes())
        f.write(d.x.numpy().astype("<f4").tobytes())
        f.write(d.edge_index.contiguous().numpy().astype("<i8").tobytes())
        f.write(d.edge_attr.numpy().astype("<f4").tobytes())
    model = model.cuda()
    with torch.no_grad():
        want, want_h = model(type("D", (), dict(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda()))())
    # without a weight-plane cache (the split-K kernels) and with one (ABI v6: the few-row kernels the module runs too)
    for extra, tol in (([], 1e-5), (["cache"], 2e-6)):
        out = str(tmp_path / "out.bin")
        r = subprocess.run([exe, blob, out] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = np.fromfile(out, dtype="<f4")
        logits = torch.from_numpy(raw[:2 * e * 2].reshape(2, e, 2))
        h = torch.from_numpy(raw[2 * e * 2:].reshape(n, 32))
        for i in range(2):
            assert (logits[i] - want["classified_edges"][i].cpu()).abs().max().item() <= tol, extra
        assert (h - want_h.cpu()).abs().max().item() <= 1e-5 * max(1.0, want_h.abs().max().item())
