"""Shared helpers for the MPN golden fixtures (tests/golden/g*.npz, made by tests/golden/make_golden.py)."""
import copy
import glob
import hashlib
import json
import os

import numpy as np
import torch

import mtmc_mpn
from mtmc_mpn import graphs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARCH = "resnet101"


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


class Case:
    def __init__(self, name):
        self.name = name
        self.blob = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(self.blob["meta"]))
        self.sub_idx = torch.from_numpy(self.blob["sub_idx"])

    def params(self):
        p = copy.deepcopy(mtmc_mpn.DEFAULT_GRAPH_NET_PARAMS)
        for k, v in self.meta["overrides"].items():
            node, keys = p, k.split(".")
            for kk in keys[:-1]:
                node = node[kk]
            node[keys[-1]] = v
        return p

    def model(self):
        """Seeded construction == the reference's initial weights (checked against param_sha)."""
        torch.manual_seed(self.meta["weight_seed"])
        m = mtmc_mpn.MOTMPNet(self.params(), None, ARCH).eval()
        for k, v in m.state_dict().items():
            assert sha(v) == self.meta["param_sha"][k], f"{self.name}: weight {k} differs from the reference's"
        return m

    def graph(self):
        r = self.meta["recipe"]
        if r["kind"] == "random":
            d = graphs.random_graph(r["n"], r["e"], 2048, r["seed"])
        elif r["kind"] == "cams":
            d = graphs.camera_graph(tuple(r["cams"]), 2048, r["seed"])
        elif r["kind"] == "train":
            with open(os.path.join(GOLDEN_DIR, "train_tracklets.json")) as f:
                d = graphs.training_graph(json.load(f)["tracklets"], r["n_ids"], 2048, r["seed"])
        else:
            raise ValueError(r["kind"])
        if self.meta["perm_seed"] is not None:
            perm = torch.randperm(d.edge_index.shape[1], generator=torch.Generator().manual_seed(self.meta["perm_seed"]))
            d.edge_index = d.edge_index[:, perm]
            d.edge_attr = d.edge_attr[perm]
        return d

    def inputs_match_reference_run(self, d):
        s = self.meta["input_sha"]
        return sha(d.x) == s["x"] and sha(d.edge_index) == s["edge_index"] and sha(d.edge_attr) == s["edge_attr"]

    def logits(self, i, f64=False):
        return torch.from_numpy(self.blob[("logits64_%d" if f64 else "logits_%d") % i])

    def h(self, f64=False):
        return torch.from_numpy(self.blob["h64" if f64 else "h"])
