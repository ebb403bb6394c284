"""Seeded inputs of the graph-construction / feature-reader fixtures (tests/golden/gb_*.npz, fs_reader.npz), shared
by the generator (tests/golden/make_golden_graph.py, which runs the reference on them) and the tests."""
import json
import os
import pickle

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S02_GT_CAMS = (124, 90, 99, 137)          # tracklets per camera c006..c009 in the reference's eval/ground_truth_S02.txt

CASES = ("s02_gt", "cams3", "interleaved")


def inputs(name):
    """(features [N,2048] f32, camera id per node, identity per node) in the order the dataset yields tracklets."""
    if name == "s02_gt":          # camera by camera (libs/dataset.py:279-281), 145 identities spread over the cameras
        cams = np.repeat(np.array([6, 7, 8, 9]), S02_GT_CAMS)
        ids = torch.randint(0, 145, (cams.size,), generator=torch.Generator().manual_seed(3)).numpy()
        feats = torch.randn(cams.size, 2048, generator=torch.Generator().manual_seed(2))
    elif name == "cams3":         # hand-checkable: cameras (3, 2, 4) -> N = 9, E = 52
        cams = np.repeat(np.array([1, 2, 3]), (3, 2, 4))
        ids = np.array([10, 11, 12, 10, 12, 10, 11, 12, 13])
        feats = torch.randn(9, 2048, generator=torch.Generator().manual_seed(7))
    elif name == "interleaved":   # training-style order: identity by identity, cameras interleave (train.py:295-302),
        with open(os.path.join(GOLDEN_DIR, "train_tracklets.json")) as f:    # near-duplicate features per identity
            tr = json.load(f)["tracklets"]
        pick = sorted({t[1] for t in tr})[:40]
        nodes = [(c, i) for i in pick for (c, j) in sorted(tr) if j == i]
        cams, ids = np.array([c for c, _ in nodes]), np.array([i for _, i in nodes])
        gen = torch.Generator().manual_seed(5)
        base = {i: torch.randn(2048, generator=gen) for i in pick}
        feats = torch.stack([base[i] + 0.05 * torch.randn(2048, generator=gen) for _, i in nodes])
    else:
        raise KeyError(name)
    return feats, cams, ids


def feature_scene(seed=0, n_per_cam=(7, 5, 9), f=2048):
    g = torch.Generator().manual_seed(seed)
    cams, ids = [], []
    for c, n in zip((6, 7, 9), n_per_cam):
        cams += [c] * n
        ids += sorted(torch.randperm(400, generator=g)[:n].tolist())
    return np.asarray(cams), np.asarray(ids), torch.randn(len(cams), f, generator=g)


def dump_reference_layout(root, scene, file, model, cams, ids, feats):
    """Pickles exactly as libs/reid_feature_extraction.py:165-182 writes them (one CPU tensor per tracklet)."""
    for c, i, t in zip(cams, ids, feats):
        d = os.path.join(root, scene, "c" + str(int(c)).zfill(3), str(int(i)).zfill(4))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, file + "_" + model + ".pkl"), "wb") as fout:
            pickle.dump(t.cpu().clone(), fout, protocol=pickle.HIGHEST_PROTOCOL)
