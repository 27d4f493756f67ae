"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for the compiled nms.c):

    python tests/golden/make_golden.py

nms_cases.npz   inputs + outputs of the reference's OWN nms.c (compiled unmodified, oracle/_ref/libnms_ref.so):
                NMS keep tables and bbox_vote results for the SURVEY §8d score regimes, plus the IoU known-answer
                vector of test.lua:40-52.  These pin both the C restatement and the HIP kernels.
modules.npz     small inputs/outputs of the Lua-source-pinned modules as restated by the oracle (regression pin).
frcnn_small.npz a tiny VGG-shaped Fast R-CNN image: inputs, weights seed and the oracle's scores / boxes.
nms_dense.npz   (round 3; `python tests/golden/make_golden.py nms_dense` writes only this file) utils.nms_dense (utils.lua:402-462) as
                restated by the oracle — a REGRESSION PIN of the restatement, not a reference vector (no Lua runtime exists here; the
                order among bit-equal scores is the oracle's stable sort, unpinned against TH's quicksort).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scored_boxes  # noqa: E402
from oracle import mpn_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make_nms_dense():
    out = {}
    rng = np.random.default_rng(20260924)
    i = 0
    for regime in ("distinct", "ties", "saturated"):
        for n in (1, 7, 64, 65, 300, 1000):
            sb = random_scored_boxes(rng, n, regime, span=300.0 if n <= 65 else 1000.0)
            for thr in (0.3, 0.5):
                out["d%d_in" % i], out["d%d_thr" % i], out["d%d_pick" % i] = sb, np.float32(thr), O.nms_dense(sb, thr).astype(np.int32)
                i += 1
    out["n_cases"] = np.int32(i)
    np.savez_compressed(os.path.join(HERE, "nms_dense.npz"), **out)


def main():
    if sys.argv[1:] == ["nms_dense"]:
        make_nms_dense()
        print("wrote nms_dense.npz")
        return
    make_nms_dense()
    assert O.have_ref(), "needs oracle/_ref/libnms_ref.so (make -C oracle ref with /root/reference present)"
    out = {}
    rng = np.random.default_rng(20260923)
    i = 0
    for regime in ("distinct", "ties", "saturated", "allequal"):
        for n in (1, 7, 64, 65, 300, 1000):
            sb = random_scored_boxes(rng, n, regime, span=300.0 if n <= 65 else 1000.0)
            sb[:, 4] = np.maximum(sb[:, 4], 1e-3)
            for thr in (0.3, 0.5):
                keep = O.ref_nms(sb, thr)
                out["nms%d_in" % i] = sb
                out["nms%d_thr" % i] = np.float32(thr)
                out["nms%d_keep" % i] = keep
                out["nms%d_vote" % i] = O.ref_bbox_vote(keep, sb, 0.5)
                i += 1
    out["n_cases"] = np.int32(i)
    a = np.array([[0, 0, 100, 100], [0, 50, 100, 150], [50, 0, 150, 100], [50, 50, 150, 150], [100, 100, 200, 200]], np.float32)
    out["iou_a"], out["iou_b"] = a, np.array([50, 50, 150, 150], np.float32)
    out["iou_ref"] = np.array([O.ref_overlap(r, [50, 50, 150, 150]) for r in a], np.float32)
    out["iou_lua_gt"] = np.array([1 / 7, 1 / 3, 1 / 3, 1, 1 / 7], np.float32)  # test.lua:49
    np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **out)

    rng = np.random.default_rng(7)
    rois = np.concatenate([np.ones((40, 1)), rng.uniform(1, 500, (40, 2)), rng.uniform(501, 999, (40, 2))], 1).astype(np.float32)
    feat = rng.standard_normal((1, 16, 38, 63)).astype(np.float32)
    pooled, arg = O.roi_pool(feat, rois * np.array([1, 1, 0.6, 1, 0.6], np.float32), 7, 7, 1 / 16)
    d = (rng.standard_normal((40, 12)) * 0.2).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "modules.npz"), rois=rois, foveal=O.foveal(rois), ctx15=O.context_region(rois, 1.5),
                        feat=feat, roi_scale=np.array([1, 1, 0.6, 1, 0.6], np.float32), pooled=pooled, argmax=arg, deltas=d,
                        decoded=O.bbox_decode(rois[:, 1:], d), bbox_norm=O.bbox_norm(d, [0, 0.01, -0.02, 0.03], [0.1, 0.1, 0.2, 0.2]))

    from multipathnet_amd import models
    cfg = [8, 16, "P", 16, "P", 32]
    P = models.synthetic_params(cfg, pooled=7, fc_dim=64, n_classes=5, seed=557)
    Pn = {k: ([t.numpy() for t in v] if isinstance(v, list) and v and hasattr(v[0], "numpy") else (v.numpy() if hasattr(v, "numpy") else v))
          for k, v in P.items()}
    rng = np.random.default_rng(555)
    H, W, N = 75, 125, 24
    im = rng.random((3, H, W), dtype=np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(8), np.log(60), (N, 2)))
    boxes = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)
    feat = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg)
    logits, deltas = O.frcnn_head(feat, O.project_im_rois(boxes, 1.0), Pn, pooled=7, spatial_scale=0.25)
    np.savez_compressed(os.path.join(HERE, "frcnn_small.npz"), image=im, boxes=boxes, conv5=feat, scores=O.softmax(logits),
                        bbox=O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H), seed=np.int32(557))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
