"""Committed golden vectors (tests/golden/, made by make_golden.py in the build container from the reference's own
compiled nms.c and from the oracle).  CPU: the C restatement reproduces them; GPU: the HIP kernels reproduce them."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def nms_cases():
    z = np.load(os.path.join(G, "nms_cases.npz"))
    return z, int(z["n_cases"])


def test_oracle_reproduces_reference_nms_goldens(O, nms_cases):
    z, n = nms_cases
    assert n == 48
    for i in range(n):
        sb, thr = z["nms%d_in" % i], float(z["nms%d_thr" % i])
        keep = O.nms(sb, thr)
        assert np.array_equal(keep, z["nms%d_keep" % i]), i
        assert np.array_equal(O.bbox_vote(keep, sb, 0.5), z["nms%d_vote" % i], equal_nan=True), i
    assert np.array_equal(O.boxoverlap(z["iou_a"], z["iou_b"]), z["iou_ref"])
    assert np.abs(z["iou_ref"] - z["iou_lua_gt"]).max() < 5e-3  # test.lua:51


def test_oracle_reproduces_module_goldens(O):
    z = np.load(os.path.join(G, "modules.npz"))
    assert np.array_equal(O.foveal(z["rois"]), z["foveal"])
    assert np.array_equal(O.context_region(z["rois"], 1.5), z["ctx15"])
    p, a = O.roi_pool(z["feat"], z["rois"] * z["roi_scale"], 7, 7, 1 / 16)
    assert np.array_equal(p, z["pooled"]) and np.array_equal(a, z["argmax"])
    assert np.array_equal(O.bbox_decode(z["rois"][:, 1:], z["deltas"]), z["decoded"])
    assert np.array_equal(O.bbox_norm(z["deltas"], [0, 0.01, -0.02, 0.03], [0.1, 0.1, 0.2, 0.2]), z["bbox_norm"])


@pytest.mark.gpu
def test_hip_reproduces_reference_nms_goldens(dev, nms_cases):
    from multipathnet_amd import utils
    z, n = nms_cases
    for i in range(n):
        sb, thr = z["nms%d_in" % i], float(z["nms%d_thr" % i])
        d = torch.from_numpy(sb).to(dev)
        keep = utils.nms(d, thr)
        assert np.array_equal(keep.cpu().numpy(), z["nms%d_keep" % i]), i
        vote = utils.bbox_vote(keep.contiguous(), d, 0.5)
        assert np.array_equal(vote.cpu().numpy(), z["nms%d_vote" % i], equal_nan=True), i


@pytest.mark.gpu
def test_hip_reproduces_module_and_pipeline_goldens(dev):
    from multipathnet_amd import nn, utils, models
    z = np.load(os.path.join(G, "modules.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    assert np.array_equal(nn.Foveal().forward(t(z["rois"])).cpu().numpy(), z["foveal"])
    assert np.array_equal(nn.ContextRegion(1.5).forward(t(z["rois"])).cpu().numpy(), z["ctx15"])
    m = nn.ROIPooling(7, 7, 1 / 16)
    assert np.array_equal(m.forward([t(z["feat"]), t(z["rois"] * z["roi_scale"])]).cpu().numpy(), z["pooled"])
    assert np.array_equal(m.indices.cpu().numpy(), z["argmax"])
    assert np.abs(utils.decode_all_classes(t(z["rois"][:, 1:].copy()), t(z["deltas"])).cpu().numpy() - z["decoded"]).max() < 1e-3
    f = np.load(os.path.join(G, "frcnn_small.npz"))
    cfg = [8, 16, "P", 16, "P", 32]
    P = models.synthetic_params(cfg, pooled=7, fc_dim=64, n_classes=5, seed=int(f["seed"]))
    H, W = f["image"].shape[1:]
    net = models.FastRCNN(P, cfg=cfg, pooled=7, spatial_scale=0.25, max_h=H, max_w=W, max_rois=f["boxes"].shape[0])
    scores, bbox = net.detect(t(f["image"]), t(f["boxes"]))
    assert np.abs(scores.cpu().numpy() - f["scores"]).max() < 1e-4
    assert np.abs(bbox.cpu().numpy() - f["bbox"]).max() < 1e-4 * W
    conv5 = net.debug_tensor("conv5", f["conv5"].shape).cpu().numpy()
    assert np.abs(conv5 - f["conv5"]).max() < 1e-4 * max(1.0, np.abs(f["conv5"]).max())


def test_oracle_reproduces_nms_dense_pins(O):
    """utils.nms_dense: oracle-generated regression pins (tests/golden/nms_dense.npz; not reference vectors — see make_golden.py)"""
    z = np.load(os.path.join(G, "nms_dense.npz"))
    n = int(z["n_cases"])
    assert n == 36
    for i in range(n):
        assert np.array_equal(O.nms_dense(z["d%d_in" % i], float(z["d%d_thr" % i])), z["d%d_pick" % i]), i


@pytest.mark.gpu
def test_hip_reproduces_nms_dense_pins(dev):
    from multipathnet_amd import utils
    z = np.load(os.path.join(G, "nms_dense.npz"))
    for i in range(int(z["n_cases"])):
        got = utils.nms_dense(torch.from_numpy(z["d%d_in" % i]).to(dev), float(z["d%d_thr" % i]))
        assert np.array_equal(got.cpu().numpy(), z["d%d_pick" % i]), i
