"""Scene feature blob: round trip from the reference's per-tracklet pickle layout, bit for bit."""
import os
import pickle

import numpy as np
import pytest
import torch

from mtmc_mpn import feature_store as fs


def _dump_reference_layout(root, scene, file, model, cams, ids, feats):
    """Write pickles exactly like libs/reid_feature_extraction.py:165-182 does (one CPU tensor per tracklet)."""
    for c, i, t in zip(cams, ids, feats):
        d = os.path.join(root, scene, "c" + str(int(c)).zfill(3), str(int(i)).zfill(4))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, file + "_" + model + ".pkl"), "wb") as fout:
            pickle.dump(t.cpu(), fout, protocol=pickle.HIGHEST_PROTOCOL)


def _scene(seed=0, n_per_cam=(7, 5, 9), f=2048):
    g = torch.Generator().manual_seed(seed)
    cams, ids = [], []
    for c, n in zip((6, 7, 9), n_per_cam):
        cams += [c] * n
        ids += sorted(torch.randperm(400, generator=g)[:n].tolist())
    feats = torch.randn(len(cams), f, generator=g)
    return np.asarray(cams), np.asarray(ids), feats


def test_convert_round_trip(tmp_path):
    cams, ids, feats = _scene()
    perm = np.random.RandomState(0).permutation(len(cams))          # directory creation order must not matter
    _dump_reference_layout(str(tmp_path / "reid_features"), "S02", "mtsc_x", "resnet101",
                           cams[perm], ids[perm], feats[torch.from_numpy(perm)])
    blob = str(tmp_path / "S02.feat")
    assert fs.convert(str(tmp_path / "reid_features"), "S02", "mtsc_x", "resnet101", blob) == (len(cams), 2048)
    st = fs.FeatureStore(blob)
    assert len(st) == len(cams) and st.f == 2048
    c2, i2 = st.tracklets()
    assert np.array_equal(c2, cams) and np.array_equal(i2, ids)     # (camera, id) order = the dataset's order
    assert np.array_equal(np.asarray(st.feats), feats.numpy())      # bit for bit
    rows = st.rows([9, 6], [ids[-1], ids[0]])
    assert rows.tolist() == [len(cams) - 1, 0]
    assert os.path.getsize(blob) % 4 == 0 and st.feats.offset % 4096 == 0
    x = st.to_device("cpu", cams=[7, 6], ids=[ids[7], ids[1]])
    assert torch.equal(x, feats[[7, 1]])


def test_errors(tmp_path):
    cams, ids, feats = _scene(1, (3, 2, 2), 64)
    blob = str(tmp_path / "a.feat")
    fs.write(blob, cams, ids, feats)
    st = fs.FeatureStore(blob)
    with pytest.raises(KeyError):
        st.rows([6], [9999])
    with pytest.raises(ValueError, match="duplicate"):
        fs.write(blob, [1, 1], [5, 5], torch.zeros(2, 64))
    data = open(blob, "rb").read()
    open(str(tmp_path / "t.feat"), "wb").write(data[:-8])
    with pytest.raises(ValueError, match="truncated"):
        fs.FeatureStore(str(tmp_path / "t.feat"))
    open(str(tmp_path / "m.feat"), "wb").write(b"X" + data[1:])
    with pytest.raises(ValueError, match="magic"):
        fs.FeatureStore(str(tmp_path / "m.feat"))
    with pytest.raises(FileNotFoundError):
        os.makedirs(str(tmp_path / "empty" / "S01"))
        fs.convert(str(tmp_path / "empty"), "S01", "f", "m", blob)


@pytest.mark.gpu
def test_blob_to_graph_on_gpu(tmp_path):
    """One H2D copy of the mapped matrix feeds build_graph; same graph as from the in-memory features."""
    import mtmc_mpn
    cams, ids, feats = _scene(2, (20, 15, 25))
    blob = str(tmp_path / "s.feat")
    fs.write(blob, cams, ids, feats)
    st = fs.FeatureStore(blob)
    x = st.to_device("cuda:0", pin=True)
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), feats)
    c, i = st.tracklets()
    g1 = mtmc_mpn.build_graph(x, c, i)
    g2 = mtmc_mpn.build_graph(feats.to("cuda:0"), cams, ids)
    assert torch.equal(g1.edge_index, g2.edge_index) and torch.equal(g1.edge_attr, g2.edge_attr)
    assert torch.equal(g1.x, g2.x)
