"""inn.ROIPooling mirror vs the oracle: pooled values AND argmax indices bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("PH,PW,scale", [(7, 7, 1 / 16), (6, 6, 1 / 16), (14, 14, 1 / 16), (7, 7, 1 / 8), (7, 7, 1 / 4), (17, 17, 17 / 299)])
def test_roi_pool_bit_exact(O, dev, PH, PW, scale):
    from multipathnet_amd import nn
    rng = np.random.default_rng(int(PH * 100 + 1 / scale))
    feat = rng.standard_normal((2, 24, 38, 63)).astype(np.float32)
    n = 200
    c = rng.uniform([1, 1], [1000, 600], (n, 2))
    wh = np.exp(rng.uniform(np.log(2), np.log(700), (n, 2)))
    rois = np.concatenate([rng.integers(1, 3, (n, 1)), c - wh / 2, c + wh / 2], 1).astype(np.float32)
    m = nn.ROIPooling(PW, PH, scale)
    out = m.forward([_t(feat, dev), _t(rois, dev)])
    ref, arg = O.roi_pool(feat, rois, PH, PW, scale)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(m.indices.cpu().numpy(), arg)


def test_roi_pool_reference_test_inputs(O, dev):
    """test.lua:141-146: 1x512x38x50 map, 40 ROIs = randn*50 (negative / inverted boxes), batch idx 1;
    and chunked (25) == un-chunked exactly (test.lua:150-162)."""
    from multipathnet_amd import nn
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((1, 512, 38, 50)).astype(np.float32)
    rois = (rng.standard_normal((40, 5)) * 50).astype(np.float32)
    rois[:, 0] = 1
    m = nn.ROIPooling(7, 7, 1 / 16)
    full = m.forward([_t(feat, dev), _t(rois, dev)]).clone()
    ref, _ = O.roi_pool(feat, rois, 7, 7, 1 / 16)
    assert np.array_equal(full.cpu().numpy(), ref)
    parts = torch.cat([m.forward([_t(feat, dev), _t(rois[:25], dev)]).clone(), m.forward([_t(feat, dev), _t(rois[25:], dev)]).clone()])
    assert torch.equal(full, parts)
    assert m.forward([_t(feat, dev), torch.zeros((0, 5), device=dev)]).shape == (0, 512, 7, 7)


def test_roi_pool_convention_is_parameterised(O, dev):
    from multipathnet_amd import nn
    rng = np.random.default_rng(1)
    feat = rng.standard_normal((1, 8, 38, 63)).astype(np.float32)
    rois = np.array([[1, 17, 33, 400, 300], [1, 1, 1, 1000, 600]], np.float32)
    for off, adj in [(0.0, 0), (1.0, -1), (1.0, 0)]:
        m = nn.ROIPooling(7, 7, 1 / 16, coord_offset=off, end_adjust=adj)
        ref, arg = O.roi_pool(feat, rois, 7, 7, 1 / 16, coord_offset=off, end_adjust=adj)
        assert np.array_equal(m.forward([_t(feat, dev), _t(rois, dev)]).cpu().numpy(), ref)
        assert np.array_equal(m.indices.cpu().numpy(), arg)


@pytest.mark.parametrize("C,H,W,N,scale", [(16, 150, 250, 300, 0.25), (24, 75, 125, 200, 0.125), (8, 38, 63, 128, 0.0625), (8, 1, 1, 5, 1.0),
                                           (8, 2, 37, 40, 0.5), (40, 9, 3, 33, 0.1)])
def test_roi_pool_range_max_tables_equal_direct(dev, C, H, W, N, scale):
    """the MultiPathNet head's ROI pool reads vertical range-max tables (2 x bin-width reads per bin): its output must be
    identical to the direct kernel's — signed features, regions up to 4x the image (Foveal), degenerate and outside boxes"""
    import ctypes
    from multipathnet_amd import _lib
    lib = _lib.load("debug")  # the comparison helper is a test hook: libmpn_hip_dbg.so only (same kernels as the product library)
    rng = np.random.default_rng(C * 100 + H)
    feat = rng.standard_normal((C, H, W)).astype(np.float32)
    img_w, img_h = W / scale, H / scale
    cx, cy = rng.uniform(-0.2, 1.2, N) * img_w, rng.uniform(-0.2, 1.2, N) * img_h
    bw, bh = np.exp(rng.uniform(np.log(1.0), np.log(4 * img_w), N)), np.exp(rng.uniform(np.log(1.0), np.log(4 * img_h), N))
    rois = np.stack([np.ones(N), cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
    rois[0, 1:] = [5, 5, 5, 5]                      # 1-pixel box
    rois[1 % N, 1:] = [-500, -500, -400, -400]      # entirely outside -> empty bins -> 0
    f = torch.from_numpy(feat).to(dev)
    r = torch.from_numpy(rois).to(dev)
    n = ctypes.c_int(-1)
    rc = lib.mpn_debug_roi_pool_rmq_mismatches(ctypes.c_void_p(f.data_ptr()), C, H, W, ctypes.c_void_p(r.data_ptr()), 5, N, 7, 7,
                                               ctypes.c_float(scale), ctypes.byref(n))
    assert rc == 0 and n.value == 0


# ---- inn.ROIPooling's CPU branch: clip first, then nn.SpatialAdaptiveMaxPooling's bins (VERDICT r5 task 4a; /root/reference/models/alexnet.lua:23) ----
def _fov_rois(O, rng, n, img_w, img_h):
    """in-image boxes, their four Foveal regions (x1.5 / x2 / x4 leave the image), boxes hanging over every border, degenerate ones"""
    c = rng.uniform([1, 1], [img_w, img_h], (n, 2))
    wh = np.exp(rng.uniform(np.log(2), np.log(0.7 * img_w), (n, 2)))
    base = np.concatenate([np.ones((n, 1)), np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [img_w, img_h, img_w, img_h])], 1).astype(np.float32)
    over = np.concatenate([np.ones((n, 1)), c - wh, c + wh], 1).astype(np.float32)
    rois = np.concatenate([base, O.foveal(base), over], 0)
    rois[0, 1:] = [5, 5, 5, 5]
    rois[1, 1:] = [-900, -700, -800, -600]
    rois[2, 1:] = [img_w + 50, img_h + 50, img_w + 300, img_h + 200]
    return rois


@pytest.mark.parametrize("PH,PW,scale,H,W", [(7, 7, 1 / 16, 38, 63), (6, 6, 1 / 16, 38, 63), (14, 14, 1 / 16, 38, 63), (7, 7, 1 / 8, 75, 125),
                                             (7, 7, 1 / 4, 150, 250), (17, 17, 17 / 299, 35, 60)])
def test_roi_pool_adaptive_rule_bit_exact(O, dev, PH, PW, scale, H, W):
    """values AND arg-max cells of the module-level op under MPN_ROI_BINS_ADAPTIVE == the oracle's crop + adaptive-max-pool restatement
    (which tests/test_oracle_roipool_adaptive.py pins to PyTorch's adaptive_max_pool2d), incl. Foveal regions that leave the image"""
    from multipathnet_amd import nn
    rng = np.random.default_rng(int(PH * 100 + H))
    feat = rng.standard_normal((2, 24, H, W)).astype(np.float32)
    rois = _fov_rois(O, rng, 60, W / scale, H / scale)
    rois[:, 0] = rng.integers(1, 3, rois.shape[0])
    m = nn.ROIPooling(PW, PH, scale, bin_rule=nn.ROIPooling.BINS_ADAPTIVE)
    out = m.forward([_t(feat, dev), _t(rois, dev)])
    ref, arg = O.roi_pool(feat, rois, PH, PW, scale, bin_rule=O.ROI_BINS_ADAPTIVE)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(m.indices.cpu().numpy(), arg)
    assert not np.isinf(ref).any() and (arg >= 0).all()            # the adaptive bins are never empty
    ref0, _ = O.roi_pool(feat, rois, PH, PW, scale, bin_rule=O.ROI_BINS_CAFFE)
    assert not np.array_equal(ref, ref0)                           # and it IS a different rule on these windows
    m0 = nn.ROIPooling(PW, PH, scale)                              # the default stays the CUDA branch
    assert np.array_equal(m0.forward([_t(feat, dev), _t(rois, dev)]).cpu().numpy(), ref0)


@pytest.mark.parametrize("C,H,W,N,scale", [(16, 150, 250, 300, 0.25), (24, 75, 125, 200, 0.125), (8, 38, 63, 128, 0.0625), (8, 1, 1, 5, 1.0),
                                           (8, 2, 37, 40, 0.5), (40, 9, 3, 33, 0.1)])
def test_roi_pool_range_max_tables_equal_direct_adaptive_rule(dev, C, H, W, N, scale):
    """the pipeline's pooling kernels (C8P direct, C8P range-max tables, pixel-major range-max tables) agree bit for bit under the adaptive
    rule too (same shared bin arithmetic: mpn_internal.h roi_bin_bounds)"""
    import ctypes
    from multipathnet_amd import _lib
    lib = _lib.load("debug")
    rng = np.random.default_rng(C * 100 + H + 1)
    feat = rng.standard_normal((C, H, W)).astype(np.float32)
    img_w, img_h = W / scale, H / scale
    cx, cy = rng.uniform(-0.2, 1.2, N) * img_w, rng.uniform(-0.2, 1.2, N) * img_h
    bw, bh = np.exp(rng.uniform(np.log(1.0), np.log(4 * img_w), N)), np.exp(rng.uniform(np.log(1.0), np.log(4 * img_h), N))
    rois = np.stack([np.ones(N), cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
    f, r = torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev)
    n = ctypes.c_int(-1)
    lib.mpn_debug_set_roi_bins(1)
    try:
        rc = lib.mpn_debug_roi_pool_rmq_mismatches(ctypes.c_void_p(f.data_ptr()), C, H, W, ctypes.c_void_p(r.data_ptr()), 5, N, 7, 7,
                                                   ctypes.c_float(scale), ctypes.byref(n))
    finally:
        lib.mpn_debug_set_roi_bins(0)
    assert rc == 0 and n.value == 0


def test_pipelines_with_the_adaptive_rule_vs_oracle(O, dev):
    """mpn_frcnn_config.roi_bin_rule = MPN_ROI_BINS_ADAPTIVE through the fused pipelines: VGG-shaped Fast R-CNN (pixel-major pooling into fc6's
    operand: the pooled tensor bit for bit, scores / boxes within 1e-4), the MultiPathNet head (Foveal regions x range-max-table pooling of three
    maps) and an AlexNet-shaped op-list graph (BASELINE configs[0]'s "CPU nn path": 6 x 6 pooling into the fully-connected head) — each
    against the oracle's whole-model restatement pooling with the same rule, and each different from the default rule's result."""
    from multipathnet_amd import models
    from test_gpu_pipeline import SMALL, _boxes, _np_params, _np_tree
    s = SMALL
    rng = np.random.default_rng(77)
    im = rng.random((3, s["H"], s["W"]), dtype=np.float32)
    boxes = _boxes(np.random.default_rng(78), s["N"], s["W"], s["H"])
    P = models.synthetic_params(s["cfg"], pooled=7, fc_dim=s["fc"], n_classes=s["C"], seed=557)
    Pn = _np_params(P)
    feat = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], s["cfg"])
    rois = O.project_im_rois(boxes, 1.0)
    got = {}
    for rule in (1, 0):
        net = models.FastRCNN(P, cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"], roi_bin_rule=rule)
        scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
        torch.cuda.synchronize()
        with O.roi_bin_rule(rule):
            pooled, _ = O.roi_pool(feat, rois, 7, 7, s["scale"])
            logits, deltas = O.frcnn_head(feat, rois, Pn, pooled=7, spatial_scale=s["scale"], chunk=500)
        dp = net.debug_tensor("pooled", pooled.shape).cpu().numpy()
        conv5 = net.debug_tensor("conv5", feat.shape).cpu().numpy()
        # the device pools ITS conv5 (within 1e-4 of the oracle's): compare the pooling itself on the device's own map, bit for bit
        with O.roi_bin_rule(rule):
            pooled_dev, _ = O.roi_pool(conv5, rois, 7, 7, s["scale"])
        assert np.array_equal(dp, pooled_dev), rule
        assert np.abs(scores.cpu().numpy() - O.softmax(logits)).max() < 1e-4
        assert np.abs(bbox.cpu().numpy() - O.clamp_boxes(O.bbox_decode(boxes, deltas), s["W"], s["H"])).max() < 1e-4 * s["W"]
        got[rule] = dp
    assert not np.array_equal(got[0], got[1])
    # MultiPathNet head
    cfg = [8, 16, "P", 16, 24, "P", 32, 32, "P", 64, "P", 64]
    H, W, N, Cn, K = 150, 250, 120, 9, 3
    Pm = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=11)
    rng = np.random.default_rng(21)
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, N, W, H, lo=12)
    Pmn = _np_tree(Pm)
    taps = {}
    O.vgg_trunk(O.image_transform(im, **O.ROSS), Pmn["conv_w"], Pmn["conv_b"], cfg, taps=taps)
    outs = {}
    for rule in (1, 0):
        net = models.MultiPathNet(Pm, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N, roi_bin_rule=rule)
        scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
        torch.cuda.synchronize()
        with O.roi_bin_rule(rule):
            ref_scores, deltas = O.mpnet_head([taps["conv5"], taps["conv4"], taps["conv3"]], O.project_im_rois(boxes, 1.0), Pmn)
        assert np.abs(scores.cpu().numpy() - ref_scores).max() < 1e-4, rule
        assert np.abs(bbox.cpu().numpy() - O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)).max() < 1e-4 * W
        outs[rule] = scores.cpu().numpy()
    assert np.abs(outs[0] - outs[1]).max() > 1e-4           # the Foveal regions leave the image: the rule matters to the scores
