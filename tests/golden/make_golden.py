#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation itself.

Runs only in the build container (needs /root/reference, which never travels): it imports the
reference's models/mpn.py + models/mlp.py *unmodified* -- with the build-owned torch_scatter
stand-in (tests/golden/_standin, the third-party op is not installed) ahead on sys.path --
builds seeded models and seeded graphs, evaluates them in fp32 and fp64 on CPU and stores
inputs-by-recipe + expected outputs as small fixtures.  Nothing of the reference's source is
copied: fixtures hold seeds, hashes and numbers only.

While generating it also asserts, for every case, that
  * mtmc_mpn.MOTMPNet built under the same seed has a bit-identical state_dict, and
  * oracle/mpn_oracle.forward reproduces the reference's fp32 outputs bit for bit,
so the committed fixtures pin both.

    python tests/golden/make_golden.py            # rewrites every fixture
"""
import copy
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MTMC_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_standin"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import mtmc_mpn  # noqa: E402
from mtmc_mpn import graphs  # noqa: E402
from oracle import mpn_oracle  # noqa: E402
from models.mpn import MOTMPNet as RefMOTMPNet  # noqa: E402  (the reference, unmodified)

ARCH = "resnet101"
SUBSET = 8192


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def ref_params(**over):
    with open(os.path.join(REF, "config", "config_training.yaml")) as f:
        p = yaml.safe_load(f)["GRAPH_NET_PARAMS"]
    assert p == mtmc_mpn.DEFAULT_GRAPH_NET_PARAMS, "shipped YAML drifted from DEFAULT_GRAPH_NET_PARAMS"
    p = copy.deepcopy(p)
    for k, v in over.items():
        node = p
        keys = k.split(".")
        for kk in keys[:-1]:
            node = node[kk]
        node[keys[-1]] = v
    return p


def build_graph(recipe):
    kind = recipe["kind"]
    if kind == "random":
        return graphs.random_graph(recipe["n"], recipe["e"], 2048, recipe["seed"])
    if kind == "cams":
        return graphs.camera_graph(tuple(recipe["cams"]), 2048, recipe["seed"])
    if kind == "train":
        with open(os.path.join(HERE, "train_tracklets.json")) as f:
            return graphs.training_graph(json.load(f)["tracklets"], recipe["n_ids"], 2048, recipe["seed"])
    raise ValueError(kind)


def subset_idx(e):
    if e <= SUBSET:
        return np.arange(e)
    return np.unique(np.linspace(0, e - 1, SUBSET).astype(np.int64))


def run_case(name, recipe, over, weight_seed=0, perm_seed=None, grads=False):
    params = ref_params(**over)
    torch.manual_seed(weight_seed)
    ref = RefMOTMPNet(copy.deepcopy(params), None, ARCH).eval()
    torch.manual_seed(weight_seed)
    mine = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, ARCH)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    assert list(sd) == list(mine.state_dict()), name
    for k, v in mine.state_dict().items():
        assert torch.equal(v, sd[k]), (name, k)

    data = build_graph(recipe)
    if perm_seed is not None:
        g = torch.Generator().manual_seed(perm_seed)
        perm = torch.randperm(data.edge_index.shape[1], generator=g)
        data.edge_index = data.edge_index[:, perm]
        data.edge_attr = data.edge_attr[perm]
    n, e = data.x.shape[0], data.edge_index.shape[1]

    with torch.no_grad():
        out32, h32 = ref(types.SimpleNamespace(x=data.x, edge_index=data.edge_index, edge_attr=data.edge_attr))
        ora32, oh32 = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, data.x, data.edge_index, data.edge_attr)
        ref64 = copy.deepcopy(ref).double()
        out64, h64 = ref64(types.SimpleNamespace(x=data.x.double(), edge_index=data.edge_index,
                                                 edge_attr=data.edge_attr.double()))
        ora64, oh64 = mpn_oracle.forward(sd, copy.deepcopy(params), ARCH, data.x, data.edge_index, data.edge_attr,
                                         dtype=torch.float64)
    assert len(out32["classified_edges"]) == len(ora32["classified_edges"])
    for a, b in zip(out32["classified_edges"], ora32["classified_edges"]):
        assert torch.equal(a, b), f"{name}: oracle fp32 != reference fp32"
    assert torch.equal(h32, oh32), name
    for a, b in zip(out64["classified_edges"], ora64["classified_edges"]):
        assert torch.equal(a, b), f"{name}: oracle fp64 != reference fp64"
    assert torch.equal(h64, oh64), name

    idx = subset_idx(e)
    blob = {"sub_idx": idx, "h": h32.numpy(), "h64": h64.numpy()}
    meta = {"name": name, "recipe": recipe, "overrides": over, "weight_seed": weight_seed,
            "perm_seed": perm_seed, "N": n, "E": e, "n_out": len(out32["classified_edges"]),
            "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(),
            "param_sha": {k: sha(v) for k, v in sd.items()},
            "input_sha": {"x": sha(data.x), "edge_index": sha(data.edge_index), "edge_attr": sha(data.edge_attr)},
            "logits_sha": [], "logits_sum": [], "logits_abssum": [], "min_margin": [], "n_pos": []}
    for i, (a, a64) in enumerate(zip(out32["classified_edges"], out64["classified_edges"])):
        blob[f"logits_{i}"] = a.numpy()[idx]
        blob[f"logits64_{i}"] = a64.numpy()[idx]
        meta["logits_sha"].append(sha(a))
        meta["logits_sum"].append(float(a.double().sum()))
        meta["logits_abssum"].append(float(a.double().abs().sum()))
        meta["min_margin"].append(float((a64[:, 1] - a64[:, 0]).abs().min()))
        meta["n_pos"].append(int((a64[:, 1] > a64[:, 0]).sum()))
    meta["h_sha"] = sha(h32)

    if grads:
        ref.train()
        labels = (torch.rand(e, generator=torch.Generator().manual_seed(77)) < 0.1).long()
        out, _ = ref(types.SimpleNamespace(x=data.x, edge_index=data.edge_index, edge_attr=data.edge_attr))
        loss = sum(torch.nn.functional.cross_entropy(o, labels) for o in out["classified_edges"])
        loss.backward()
        meta["loss"] = float(loss)
        meta["label_seed"] = 77
        meta["grad_sum"], meta["grad_abssum"] = {}, {}
        for k, p in ref.named_parameters():
            gk = p.grad.detach().reshape(-1)
            gi = subset_idx(gk.numel())
            blob["grad_idx::" + k] = gi
            blob["grad::" + k] = gk.numpy()[gi]
            meta["grad_sum"][k] = float(gk.double().sum())
            meta["grad_abssum"][k] = float(gk.double().abs().sum())

    blob["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name:28s} N={n:5d} E={e:7d} outs={meta['n_out']} n_pos={meta['n_pos']} "
          f"min|margin|={min(meta['min_margin']):.2e}  {os.path.getsize(path) / 1024:.0f} KiB")


def make_train_tracklets():
    """(camera, identity) pairs of the training scenes, derived from the reference's
    eval/ground_truth_train.txt rows `cam id frame x y w h -1 -1` (data, not code)."""
    rows = np.loadtxt(os.path.join(REF, "eval", "ground_truth_train.txt"), dtype=np.int64, usecols=(0, 1))
    pairs = sorted({(int(c), int(i)) for c, i in rows})
    with open(os.path.join(HERE, "train_tracklets.json"), "w") as f:
        json.dump({"source": "eval/ground_truth_train.txt (cam, id) uniques", "tracklets": pairs}, f)
    print("train_tracklets.json:", len(pairs), "tracklets,", len({i for _, i in pairs}), "identities")


CFG1 = {"kind": "random", "n": 64, "e": 512, "seed": 1}
S02 = {"kind": "cams", "cams": list(graphs.S02_GT_CAMS), "seed": 2}

if __name__ == "__main__":
    make_train_tracklets()
    run_case("g1_random_L1", CFG1, {})
    run_case("g2_random_L3_C3", CFG1, {"num_enc_steps": 3, "num_class_steps": 3})
    run_case("g2_random_L3_C1", CFG1, {"num_enc_steps": 3, "num_class_steps": 1})
    run_case("g3_cams324_L2", {"kind": "cams", "cams": [3, 2, 4], "seed": 3}, {"num_enc_steps": 2, "num_class_steps": 2})
    run_case("g4_s02_L1", S02, {})
    run_case("g4_s02_L3", S02, {"num_enc_steps": 3, "num_class_steps": 1})
    run_case("g5_mean", CFG1, {"num_enc_steps": 2, "num_class_steps": 2, "node_agg_fn": "mean"})
    run_case("g5_max", CFG1, {"num_enc_steps": 2, "num_class_steps": 2, "node_agg_fn": "max"})
    run_case("g5_reattach_nodes", CFG1, {"num_enc_steps": 2, "num_class_steps": 1, "reattach_initial_nodes": True})
    run_case("g5_reattach_edges", CFG1, {"num_enc_steps": 2, "num_class_steps": 1, "reattach_initial_edges": True})
    run_case("g5_reattach_both_s02", {"kind": "cams", "cams": [20, 17, 25], "seed": 5},
             {"num_enc_steps": 3, "num_class_steps": 2, "reattach_initial_nodes": True, "reattach_initial_edges": True})
    run_case("g5_L0", CFG1, {"num_enc_steps": 0, "num_class_steps": 0})
    run_case("g6_train_grads", CFG1, {
        "num_enc_steps": 3, "num_class_steps": 3,
        "encoder_feats_dict.nodes.resnet101.dropout_p": 0.0,
        "edge_model_feats_dict.dropout_p": 0.0, "node_model_feats_dict.dropout_p": 0.0}, grads=True)
    run_case("g7_s02_L3_perm", S02, {"num_enc_steps": 3, "num_class_steps": 1}, perm_seed=7)
    run_case("g8_train_topology_L3", {"kind": "train", "n_ids": 100, "seed": 3}, {"num_enc_steps": 3, "num_class_steps": 3})
