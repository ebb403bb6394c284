"""Loading of the post-processing fixtures (tests/golden/pp_*.npz: recipe + hashes + reference outputs)."""
import glob
import hashlib
import json
import os

import numpy as np
import torch

import pp_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "pp*.npz")))


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


class PpCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(z["meta"]))
        self.s = pp_cases.scenario(**self.meta["scenario"])
        assert _sha(self.s.edge_index) == self.meta["sha_edge_index"], "scenario generator drifted (edge_index)"
        assert _sha(self.s.logits) == self.meta["sha_logits"], "scenario generator drifted (logits)"
        self.flags = tuple(self.meta["flags"])
        self.prob1 = torch.from_numpy(z["prob1"])                   # the reference run's softmax(logits)[:, 1]
        self.pred_in = torch.argmax(self.s.logits, dim=1)
        self.ids_in = torch.from_numpy(z["ids_in"]).long()
        self.ids = torch.from_numpy(z["ids"]).long()
        self.pred_out = torch.zeros(self.meta["E"], dtype=torch.int64)
        self.pred_out[torch.from_numpy(z["active_out"]).long()] = 1

    def prob2(self):
        return torch.stack([1 - self.prob1, self.prob1], dim=1)
