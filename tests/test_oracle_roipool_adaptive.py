"""inn.ROIPooling's CPU branch (crop + nn.SpatialAdaptiveMaxPooling; VERDICT r5 missing #3; call sites
/root/reference/models/alexnet.lua:23, /root/reference/models/vgg.lua:28).  The oracle's orc_roi_pool_rule(bin_rule = 1) is pinned here against
an implementation that is NOT ours: PyTorch-CPU's adaptive_max_pool2d (the descendant of the THNN code the Lua module calls) applied to the
crop the module takes — values and arg-max cells, including windows that leave the map (Foveal's regions).  inn itself is absent offline, so
the corner rounding / clipping of the crop stays a restatement (parity unpinned); the POOLING of the crop is pinned."""
import numpy as np
import pytest
import torch


def _rois(rng, n, img_w, img_h, leave):
    c = rng.uniform([1, 1], [img_w, img_h], (n, 2))
    wh = np.exp(rng.uniform(np.log(2), np.log(1.2 * img_w), (n, 2)))
    r = np.concatenate([np.ones((n, 1)), c - wh / 2, c + wh / 2], 1).astype(np.float32)
    if not leave:
        r[:, 1] = np.clip(r[:, 1], 1, img_w); r[:, 3] = np.clip(r[:, 3], 1, img_w)
        r[:, 2] = np.clip(r[:, 2], 1, img_h); r[:, 4] = np.clip(r[:, 4], 1, img_h)
    return r


@pytest.mark.parametrize("PH,PW,scale,H,W", [(7, 7, 1 / 16, 38, 63), (6, 6, 1 / 16, 38, 63), (14, 14, 1 / 16, 38, 63), (7, 7, 1 / 4, 150, 250),
                                             (17, 17, 17 / 299, 35, 60), (3, 5, 1 / 8, 9, 4)])
def test_adaptive_rule_equals_pytorch_adaptive_max_pool_on_the_crop(O, PH, PW, scale, H, W):
    rng = np.random.default_rng(PH * 1000 + H)
    feat = rng.standard_normal((1, 5, H, W)).astype(np.float32)
    img_w, img_h = W / scale, H / scale
    rois = np.concatenate([_rois(rng, 60, img_w, img_h, False), _rois(rng, 60, img_w, img_h, True),
                           O.foveal(_rois(rng, 30, img_w, img_h, False))], 0)          # + the four Foveal regions of in-image boxes
    rois[0, 1:] = [5, 5, 5, 5]                                                           # a one-pixel box
    rois[1, 1:] = [-900, -700, -800, -600]                                               # entirely outside: clipped to the corner cell
    out, arg = O.roi_pool(feat, rois, PH, PW, scale, bin_rule=O.ROI_BINS_ADAPTIVE)
    ft = torch.from_numpy(feat)
    for n, r in enumerate(rois):
        c = np.round(((r[1:].astype(np.float32) - np.float32(1)) * np.float32(scale) + np.float32(1)).astype(np.float32))  # half away from zero below
        t = ((r[1:].astype(np.float32) - np.float32(1)) * np.float32(scale)).astype(np.float32) + np.float32(1)
        c = np.where(t >= 0, np.floor(t + np.float32(0.5)), -np.floor(-t + np.float32(0.5))).astype(np.int64)
        x1, y1, x2, y2 = int(np.clip(c[0], 1, W)), int(np.clip(c[1], 1, H)), int(np.clip(c[2], 1, W)), int(np.clip(c[3], 1, H))
        x2, y2 = max(x2, x1), max(y2, y1)
        crop = ft[0, :, y1 - 1:y2, x1 - 1:x2]
        ref, idx = torch.nn.functional.adaptive_max_pool2d(crop[None], (PH, PW), return_indices=True)
        assert np.array_equal(out[n], ref[0].numpy()), n
        cw = x2 - x1 + 1
        iy, ix = idx[0].numpy() // cw, idx[0].numpy() % cw
        assert np.array_equal(arg[n], (y1 - 1 + iy) * W + (x1 - 1 + ix)), n


def test_adaptive_and_cuda_branch_rules_differ_where_expected(O):
    """Where the two branches part (so that the statement in include/mpn.h is a measured one): on windows whose ROUNDED corners leave the map
    — Foveal's regions, and image-border boxes whose round((x2 - 1) * scale) is W (600 x 1000 at 1/16: column 63 of a 63-wide map) — the
    CUDA branch bins the un-clipped window and clips the bins, the CPU branch clips the window and bins the crop: nearly always different.
    On windows that stay inside they are the same bins except where the fp32 bin size rounds the other way (about one window in a thousand)."""
    rng = np.random.default_rng(11)
    H, W, scale = 38, 63, 1 / 16
    feat = rng.standard_normal((1, 4, H, W)).astype(np.float32)
    boxes = np.concatenate([_rois(rng, 1500, W / scale, H / scale, False), O.foveal(_rois(rng, 100, W / scale, H / scale, False))], 0)
    fits, diff = [], []
    for r in boxes:
        c = np.round((r[1:] - 1) * scale)
        fits.append(c[0] >= 0 and c[1] >= 0 and c[2] <= W - 1 and c[3] <= H - 1)
        diff.append(not np.array_equal(O.roi_pool(feat, r[None], 7, 7, scale, bin_rule=0)[0], O.roi_pool(feat, r[None], 7, 7, scale, bin_rule=1)[0]))
    fits, diff = np.array(fits), np.array(diff)
    print("rules differ on %d of %d windows inside the map, on %d of %d windows that leave it" % (diff[fits].sum(), fits.sum(), diff[~fits].sum(), (~fits).sum()))
    assert fits.sum() > 800 and (~fits).sum() > 300
    assert diff[fits].sum() <= 0.01 * fits.sum()
    assert diff[~fits].sum() >= 0.9 * (~fits).sum()
