"""Graph construction and the feature reader against fixtures produced by the REFERENCE's own lines
(tests/golden/make_golden_graph.py: inference.py:372-458 driven with a stub loader / stub MPN; libs/dataset.py:283-312).

CPU part: oracle/graph_oracle.py reproduces them (integers by hash, floats to host rounding) and
mtmc_mpn.feature_store returns exactly the bytes the reference's reader returned.
GPU part: `mtmc_mpn.build_graph` (csrc/graph_build.hip) against the same fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import graph_cases
from graph_cases import GOLDEN_DIR


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    blob = np.load(os.path.join(GOLDEN_DIR, f"gb_{name}.npz"), allow_pickle=False)
    return blob, json.loads(str(blob["meta"]))


@pytest.mark.parametrize("name", graph_cases.CASES)
def test_oracle_reproduces_the_reference_graph(name):
    from oracle import graph_oracle
    blob, meta = load(name)
    feats, cams, ids = graph_cases.inputs(name)
    x, ei, attr, lab = graph_oracle.build(feats, cams, ids, l2norm=True)
    assert tuple(ei.shape) == (2, meta["E"]) and x.shape[0] == meta["N"]
    assert sha(ei.contiguous().numpy()) == meta["edge_index_sha"]            # bit-exact integer work
    assert sha(lab.numpy()) == meta["edge_labels_sha"] and int(lab.sum()) == meta["positives"]
    sub, rows = torch.from_numpy(blob["sub_idx"]), torch.from_numpy(blob["row_idx"])
    assert np.abs(attr[sub].numpy() - blob["attr_sub"]).max() <= 1e-6        # host vector paths may differ in the last bit
    assert np.abs(x[rows].numpy() - blob["x_rows"]).max() <= 1e-8
    assert np.allclose(attr.double().sum(0).numpy(), meta["attr_sum"], rtol=1e-7)


def test_feature_store_returns_what_the_reference_reader_returns(tmp_path):
    from mtmc_mpn import feature_store as fs
    blob = np.load(os.path.join(GOLDEN_DIR, "fs_reader.npz"), allow_pickle=False)
    meta = json.loads(str(blob["meta"]))
    cams, ids, feats = graph_cases.feature_scene()
    assert cams.tolist() == meta["cams"] and ids.tolist() == meta["ids"]
    graph_cases.dump_reference_layout(str(tmp_path / "reid_features"), "S02", "mtsc_x", "resnet101", cams, ids, feats)
    out = str(tmp_path / "S02.feat")
    assert fs.convert(str(tmp_path / "reid_features"), "S02", "mtsc_x", "resnet101", out) == (meta["n"], meta["f"])
    st = fs.FeatureStore(out)
    got = np.asarray(st.feats)
    assert sha(got) == meta["features_sha"]               # == the bytes libs/dataset.py:298-307 unpickled, in its order
    assert np.array_equal(got[:, :8], blob["head"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", graph_cases.CASES)
def test_hip_graph_builder_against_the_reference_fixture(name):
    import mtmc_mpn
    blob, meta = load(name)
    feats, cams, ids = graph_cases.inputs(name)
    g = mtmc_mpn.build_graph(feats.cuda(), cams, ids, True)
    assert not g.edge_index.is_contiguous()                                   # the callers' [E,2].T view
    assert sha(g.edge_index.cpu().contiguous().numpy()) == meta["edge_index_sha"]
    assert sha(g.edge_labels.cpu().numpy()) == meta["edge_labels_sha"]
    sub, rows = torch.from_numpy(blob["sub_idx"]), torch.from_numpy(blob["row_idx"])
    attr, want = g.edge_attr.cpu()[sub].numpy(), blob["attr_sub"]
    # distance from the Gram form: fp32 rounding of a K=2048 dot product times the cancellation (|a|^2+|b|^2)/d^2
    assert np.abs(attr[:, 0] - want[:, 0]).max() <= 2e-5 * max(1.0, np.abs(want[:, 0]).max())
    assert np.abs(attr[:, 1] - want[:, 1]).max() <= 5e-6
    assert np.abs(g.x.cpu()[rows].numpy() - blob["x_rows"]).max() <= 1e-6 * max(1.0, np.abs(blob["x_rows"]).max())
    assert np.allclose(g.edge_attr.double().sum(0).cpu().numpy(), meta["attr_sum"], rtol=1e-5)
