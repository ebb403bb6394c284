#!/usr/bin/env python3
"""Generate tests/golden/pp_*.npz by running the REFERENCE's post-processing itself.

Runs only in the build container (needs /root/reference).  Imports the reference's inference.py and utils.py
*unmodified*; the third-party packages they import at module level but which are not installed here (cv2,
torch_geometric, torch_scatter) are satisfied by the placeholders under tests/golden/_standin (none of the
functions exercised uses cv2 / torch_geometric; scatter_add is the stand-in pinned by tests/test_oracle.py).
For every seeded scenario it runs `inference.post_processing` (inference.py:70-169) on CPU tensors, asserts that
oracle/postprocess_oracle.py reproduces ID_pred and predictions exactly, and stores recipe + hashes + outputs.

    python tests/golden/make_golden_pp.py
"""
import contextlib
import hashlib
import io
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

import mtmc_mpn  # noqa: E402,F401
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pp_cases  # noqa: E402
from oracle import postprocess_oracle as po  # noqa: E402
import inference as ref_inference  # noqa: E402  (the reference, unmodified)
import utils as ref_utils  # noqa: E402  (the reference, unmodified)

CASES = {
    # name: (scenario kwargs, flags)
    "pp1_clean": (dict(n_ids=12, n_cams=4, seed=11, fp_rate=0.0, fn_rate=0.0, sigma=0.5), (True, True, True)),
    "pp2_noisy": (dict(n_ids=30, n_cams=4, seed=12, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (True, True, True)),
    "pp3_ties": (dict(n_ids=30, n_cams=4, seed=13, fp_rate=0.01, fn_rate=0.05, pair_fp=0.015, quant=0.5),
                 (True, True, True)),
    "pp3_saturated": (dict(n_ids=25, n_cams=4, seed=14, fp_rate=0.005, fn_rate=0.05, pair_fp=0.02, quant=4.0,
                           scale=8.0), (True, True, True)),
    "pp4_perm": (dict(n_ids=30, n_cams=4, seed=15, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01, perm=True),
                 (True, True, True)),
    "pp5_cut_only": (dict(n_ids=30, n_cams=4, seed=16, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (True, False, False)),
    "pp5_prune_only": (dict(n_ids=30, n_cams=4, seed=16, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (False, True, False)),
    "pp5_split_only": (dict(n_ids=30, n_cams=4, seed=16, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (False, False, True)),
    "pp5_none": (dict(n_ids=30, n_cams=4, seed=16, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (False, False, False)),
    "pp5_prune_split": (dict(n_ids=30, n_cams=4, seed=17, fp_rate=0.01, fn_rate=0.08, pair_fp=0.01), (False, True, True)),
    "pp6_six_cams": (dict(n_ids=40, n_cams=6, seed=18, fp_rate=0.004, fn_rate=0.1, pair_fp=0.006), (True, True, True)),
    "pp7_dense": (dict(n_ids=8, n_cams=3, seed=19, fp_rate=0.5, fn_rate=0.2, pair_fp=0.3), (True, True, True)),
    "pp8_s02_scale": (dict(n_ids=140, n_cams=4, seed=20, fp_rate=0.0005, fn_rate=0.05, pair_fp=0.0006),
                      (True, True, True)),
}


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def run_reference(s, prob, pred, flags):
    edge_list = s.edge_index.cpu().numpy()
    active = [(edge_list[0][pos], edge_list[1][pos]) for pos in torch.where(pred == 1)[0]]     # inference.py:485
    ids0, _ = ref_utils.compute_SCC_and_Clusters(__import__("networkx").DiGraph(active), s.n_nodes)
    config = {"CUTTING": flags[0], "PRUNING": flags[1], "SPLITTING": flags[2]}
    data = types.SimpleNamespace(num_nodes=s.n_nodes, edge_index=s.edge_index)
    with contextlib.redirect_stdout(io.StringIO()):
        ids, predictions = ref_inference.post_processing(s.n_cams, ids0, active, pred.clone(), edge_list, config, data,
                                                         prob.clone())
    return ids0, torch.as_tensor(ids), predictions


def main():
    for name, (kw, flags) in CASES.items():
        s = pp_cases.scenario(**kw)
        prob, pred = po.classify(s.logits)
        ids0, ids_ref, pred_ref = run_reference(s, prob, pred, flags)
        ids0_o, _ = po.scc_and_clusters(po.active_edges(pred, s.edge_index.numpy()), s.n_nodes)
        assert torch.equal(ids0, ids0_o), f"{name}: oracle initial clustering != reference"
        ids_o, pred_o = po.post_processing(s.n_cams, pred, s.edge_index, s.n_nodes, prob, *flags)
        assert torch.equal(pred_ref, pred_o), f"{name}: oracle predictions != reference"
        assert torch.equal(ids_ref, ids_o), f"{name}: oracle ID_pred != reference"
        meta = {"name": name, "scenario": kw, "flags": list(flags), "N": s.n_nodes, "E": int(s.edge_index.shape[1]),
                "sha_edge_index": sha(s.edge_index), "sha_logits": sha(s.logits), "sha_prob": sha(prob),
                "active_in": int(pred.sum()), "active_out": int(pred_ref.sum()),
                "clusters_in": int(ids0.max()) + 1, "clusters_out": int(ids_ref.max()) + 1,
                "torch": torch.__version__, "networkx": __import__("networkx").__version__}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta),
                            prob1=prob[:, 1].numpy(), ids_in=ids0.numpy().astype(np.int32),
                            ids=ids_ref.numpy().astype(np.int32),
                            active_out=torch.nonzero(pred_ref == 1).flatten().numpy().astype(np.int32))
        print(f"{name}: N={s.n_nodes} E={meta['E']} active {meta['active_in']} -> {meta['active_out']}, "
              f"clusters {meta['clusters_in']} -> {meta['clusters_out']}")


if __name__ == "__main__":
    main()
