"""The whole inference step through this package: feature blob -> build_graph -> MOTMPNet -> postprocess,
against the same chain through the oracles."""
import copy

import numpy as np
import pytest
import torch

import mtmc_mpn
from mtmc_mpn import feature_store as fs
from oracle import graph_oracle, mpn_oracle, postprocess_oracle as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed,cams", [(0, (14, 11, 16)), (1, (9, 12, 10, 13))])
def test_blob_to_clusters(tmp_path, seed, cams):
    g = torch.Generator().manual_seed(seed)
    cam_ids = np.repeat(np.arange(len(cams)) + 6, cams)
    ids = np.concatenate([np.sort(torch.randperm(300, generator=g)[:n].numpy()) for n in cams])
    feats = torch.randn(len(cam_ids), 2048, generator=g)
    blob = str(tmp_path / "scene.feat")
    fs.write(blob, cam_ids, ids, feats)
    st = mtmc_mpn.FeatureStore(blob)
    c, lab = st.tracklets()
    params = mtmc_mpn.default_params(num_enc_steps=2, num_class_steps=1)
    torch.manual_seed(seed)
    model = mtmc_mpn.MOTMPNet(copy.deepcopy(params), None, "resnet101").eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # oracle chain on the CPU
    x, ei, ea, _ = graph_oracle.build(feats, cam_ids, ids)
    with torch.no_grad():
        want, want_h = mpn_oracle.forward(sd, copy.deepcopy(params), "resnet101", x, ei, ea)
    want_logits = want["classified_edges"][-1]

    # this package on the GPU
    data = mtmc_mpn.build_graph(st.to_device(DEV), c, lab)
    model = model.to(DEV)
    with torch.no_grad():
        out, h = model(data)
    logits = out["classified_edges"][-1]
    assert torch.equal(data.edge_index.cpu(), ei)
    assert (logits.cpu() - want_logits).abs().max().item() <= 1e-4
    pp = mtmc_mpn.postprocess(logits, data.edge_index, data.x.shape[0], len(cams))
    # the post-processing is exact on the numbers it was given: replay them through the oracle
    p1 = pp.preds_prob1.cpu()
    ids_o, pred_o = po.post_processing(len(cams), torch.argmax(logits.cpu(), 1), ei, len(cam_ids),
                                       torch.stack([1 - p1, p1], dim=1))
    assert torch.equal(pp.predictions.cpu(), pred_o) and torch.equal(pp.ID_pred.cpu(), ids_o)
    assert int(torch.bincount(pp.ID_pred.cpu()).max()) <= len(cams)
