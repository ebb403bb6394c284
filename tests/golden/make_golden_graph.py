#!/usr/bin/env python3
"""Generate tests/golden/gb_*.npz and fs_reader.npz by running the REFERENCE's own graph construction and feature reader.

Runs only in the build container (needs /root/reference); imports the reference's inference.py and libs/dataset.py
*unmodified* (placeholders under tests/golden/_standin for the packages that are not installed: cv2, torch_geometric --
whose `Data` is a plain attribute container --, torch_scatter, skimage, torchvision).

* gb_*: `inference.inference_precomputed_features` (inference.py:372-458) is driven with a stub loader that yields
  seeded per-tracklet features in the dataset's format (libs/dataset.py:283-312: `[{'id', 'cam': [c], 'features':
  [ndarray]}]` per tracklet) and a stub `mpn_model` that captures the `Data` object the reference hands to the MPN --
  i.e. `x`, `edge_index`, `edge_attr`, `edge_labels` come from the reference's own lines :402-456.  There is no GPU in
  this container, so `Tensor.cuda` is the identity IN THIS PROCESS ONLY.  The generator asserts that
  oracle/graph_oracle.py reproduces the captured tensors bit for bit, then stores seeds, hashes of the integer outputs
  and sampled floating-point values.
* fs_reader: pickles written the way libs/reid_feature_extraction.py:176-184 writes them are read back through the
  reference's `AIC_dataset_inference_precomputed_features.__getitem__` (libs/dataset.py:283-312; the object is created
  without running __init__, which needs the AIC detection files) and compared with mtmc_mpn.feature_store.

    python tests/golden/make_golden_graph.py
"""
import hashlib
import json
import os
import pickle
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MTMC_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_standin"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self          # no GPU here: the reference's .cuda() calls become no-ops

import graph_cases  # noqa: E402  (tests/graph_cases.py: the seeded inputs, shared with the tests)
import inference as ref_inference  # noqa: E402  (the reference, unmodified)
from libs import dataset as ref_dataset  # noqa: E402  (the reference, unmodified)
from oracle import graph_oracle  # noqa: E402


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class _Captured(Exception):
    def __init__(self, data):
        self.data = data


class _StubMPN:
    def eval(self):
        return self

    def __call__(self, data):
        raise _Captured(data)


def run_reference(feats, cams, ids):
    """The reference's loop body up to the MPN call, on one batch holding every tracklet (as main.py:80-87 makes it)."""
    batch = [{"id": ids[n], "cam": [cams[n]], "features": [feats[n].numpy()]} for n in range(len(ids))]
    reid = types.SimpleNamespace(model=types.SimpleNamespace(eval=lambda: None))
    config = {"CNN_MODEL": {"L2norm": True}, "VISUALIZE": False, "input_test": "gt"}
    try:
        ref_inference.inference_precomputed_features(config, [batch], reid, _StubMPN())
    except _Captured as c:
        return c.data
    raise AssertionError("the reference never called the MPN")


def main():
    for name in graph_cases.CASES:
        feats, cams, ids = graph_cases.inputs(name)
        data = run_reference(feats, cams, ids)
        x, ei, attr, lab = data.x, data.edge_index, data.edge_attr, data.edge_labels
        assert ei.dtype == torch.int64 and not ei.is_contiguous()          # the [E,2].T view of inference.py:413
        ox, oei, oattr, olab = graph_oracle.build(feats, cams, ids, l2norm=True)
        assert torch.equal(oei, ei) and torch.equal(olab, lab), name
        assert torch.equal(ox, x) and torch.equal(oattr, attr), name       # same ATen calls => bit-equal on this host
        e = ei.shape[1]
        sub = torch.linspace(0, e - 1, min(e, 4096)).long().unique()
        rows = torch.linspace(0, x.shape[0] - 1, min(x.shape[0], 64)).long().unique()
        meta = {"N": int(x.shape[0]), "E": int(e), "positives": int(lab.sum()), "y": [int(v) for v in data.y],
                "edge_index_sha": sha(ei.contiguous().numpy()), "edge_labels_sha": sha(lab.numpy()),
                "attr_sum": [float(v) for v in attr.double().sum(0)], "x_abs_sum": float(x.double().abs().sum())}
        np.savez_compressed(os.path.join(HERE, f"gb_{name}.npz"), meta=json.dumps(meta), sub_idx=sub.numpy(),
                            attr_sub=attr[sub].numpy(), row_idx=rows.numpy(), x_rows=x[rows].numpy(),
                            edge_index_head=ei[:, :64].contiguous().numpy())
        print(f"gb_{name}: N={meta['N']} E={meta['E']} positives={meta['positives']}")

    # ---- feature reader (libs/dataset.py:283-312) ----
    from mtmc_mpn import feature_store as fs
    cams, ids, feats = graph_cases.feature_scene()
    with tempfile.TemporaryDirectory() as tmp:
        graph_cases.dump_reference_layout(os.path.join(tmp, "reid_features"), "S02", "mtsc_x", "resnet101", cams, ids, feats)
        ds = object.__new__(ref_dataset.AIC_dataset_inference_precomputed_features)      # __init__ needs the AIC files
        ds.scenario, ds.file, ds.cnn_model_name = "S02", "mtsc_x", "resnet101"
        ds.unique_ids_all, ds.unique_cam_ids_all = ids.astype(np.float64), cams.astype(np.float64)
        ds.frames_life_ids = np.zeros((len(ids), 2))
        cwd = os.getcwd()
        os.chdir(tmp)                                        # the reader opens './reid_features/...'
        try:
            read = [ds[i][0] for i in range(len(ids))]
        finally:
            os.chdir(cwd)
        got = np.stack([r["features"][0] for r in read])
        assert [int(r["id"]) for r in read] == ids.tolist() and [int(r["cam"][0]) for r in read] == cams.tolist()
        blob = os.path.join(tmp, "S02.feat")
        fs.convert(os.path.join(tmp, "reid_features"), "S02", "mtsc_x", "resnet101", blob)
        st = fs.FeatureStore(blob)
        c2, i2 = st.tracklets()
        assert np.array_equal(c2, cams) and np.array_equal(i2, ids)
        assert np.array_equal(np.asarray(st.feats), got)                     # blob == what the reference reader returns
    np.savez_compressed(os.path.join(HERE, "fs_reader.npz"),
                        meta=json.dumps({"n": int(len(ids)), "f": int(got.shape[1]), "features_sha": sha(got),
                                         "cams": cams.tolist(), "ids": ids.tolist()}),
                        head=got[:, :8])
    print("fs_reader:", got.shape, sha(got)[:16])


if __name__ == "__main__":
    main()
