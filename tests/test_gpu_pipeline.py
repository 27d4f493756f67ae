"""End-to-end fused pipeline (mpn_frcnn_*) vs the oracle's composed per-image path, at sizes the oracle
finishes in seconds, plus size-independent properties at the BASELINE size."""
import numpy as np
import pytest
import torch

from conftest import hooks

pytestmark = pytest.mark.gpu


def _np_params(P):
    out = {}
    for k, v in P.items():
        if isinstance(v, list) and v and hasattr(v[0], "numpy"):
            out[k] = [t.numpy() for t in v]
        elif hasattr(v, "numpy"):
            out[k] = v.numpy()
        else:
            out[k] = v
    return out


def _boxes(rng, n, W, H, lo=8, hi=None):
    hi = hi or min(W, H)
    c = rng.uniform([1, 1], [W, H], (n, 2))
    wh = np.exp(rng.uniform(np.log(lo), np.log(hi), (n, 2)))
    return np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)


SMALL = dict(cfg=[8, 16, "P", 16, 24, "P", 32, 32, "P", 64, "P", 64], fc=128, C=7, H=150, W=250, N=200, scale=1 / 16)


@pytest.fixture(scope="module")
def small(O, dev):
    from multipathnet_amd import models
    s = SMALL
    P = models.synthetic_params(s["cfg"], pooled=7, fc_dim=s["fc"], n_classes=s["C"], seed=557)
    rng = np.random.default_rng(555)
    im = rng.random((3, s["H"], s["W"]), dtype=np.float32)
    boxes = _boxes(np.random.default_rng(556), s["N"], s["W"], s["H"])
    net = models.FastRCNN(P, cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"])
    Pn = _np_params(P)
    x = O.image_transform(im, **O.ROSS)
    feat = O.vgg_trunk(x, Pn["conv_w"], Pn["conv_b"], s["cfg"])
    rois = O.project_im_rois(boxes, 1.0)
    pooled, _ = O.roi_pool(feat, rois, 7, 7, s["scale"])
    logits, deltas = O.frcnn_head(feat, rois, Pn, pooled=7, spatial_scale=s["scale"], chunk=500)
    return dict(net=net, im=im, boxes=boxes, P=Pn, P_torch=P, feat=feat, pooled=pooled, logits=logits, deltas=deltas)


@pytest.mark.parametrize("fuse_pool,split,k36,pm", [(1, 0, 1, 1), (0, 0, 1, 1), (1, 2, 1, 1), (0, 3, 1, 1), (1, 0, 0, 1), (1, 0, 1, 0)])
def test_pipeline_stages_vs_oracle(O, dev, small, fuse_pool, split, k36, pm):
    from multipathnet_amd import models
    s = SMALL
    # k36 = 0: the first layer on the generic direct kernel instead of its K = 36 formulation
    # pm = 0: ROI pooling straight from the C8P map instead of from the pixel-major copy (the pooled tensor is compared bit for bit below)
    with hooks(fuse_pool=fuse_pool, conv_split=split, first_k36=k36, roi_pool_pm=pm):  # all defaults = the product library; the others build their handle on the debug flavour
        net = small["net"] if (fuse_pool, split, k36, pm) == (1, 0, 1, 1) else models.FastRCNN(
            small["P_torch"], cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"])
        scores, bbox = net.detect(torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev))
        torch.cuda.synchronize()
    feat = small["feat"]
    conv5 = net.debug_tensor("conv5", feat.shape).cpu().numpy()
    assert np.abs(conv5 - feat).max() < 1e-4 * max(1.0, np.abs(feat).max())
    pooled = net.debug_tensor("pooled", small["pooled"].shape).cpu().numpy()
    assert np.abs(pooled - small["pooled"]).max() < 1e-4 * max(1.0, np.abs(feat).max())
    cls = net.debug_tensor("cls", small["logits"].shape).cpu().numpy()
    assert np.abs(cls - small["logits"]).max() < 1e-4
    raw = net.debug_tensor("bbox_raw", small["deltas"].shape).cpu().numpy()
    assert np.abs(raw - small["deltas"]).max() < 1e-4
    ref_scores = O.softmax(small["logits"])
    ref_bbox = O.clamp_boxes(O.bbox_decode(small["boxes"], small["deltas"]), s["W"], s["H"])
    assert np.abs(scores.cpu().numpy() - ref_scores).max() < 1e-4      # north_star: class scores within 1e-4
    assert np.abs(bbox.cpu().numpy() - ref_bbox).max() < 1e-4 * s["W"]  # pixels, relative to the image extent


def test_pipeline_nms_on_own_outputs_bit_exact(O, dev, small):
    """SURVEY §7 parity protocol: kept sets are compared with the oracle fed the DEVICE's scored boxes (so
    1e-7-level score differences cannot flip a near-threshold IoU decision)."""
    s, net = SMALL, small["net"]
    im, boxes = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    dets, n = net.test_one_async(im, boxes)
    torch.cuda.synchronize()
    dets = dets[: int(n.item())].cpu().numpy()
    keep, idx, nk = [t.cpu().numpy() for t in net.nms_results()]
    scores, bbox = [t.cpu().numpy() for t in net.detect(im, boxes)]
    per = []
    for j in range(1, s["C"]):
        sb, src = O.select_scored(scores, bbox, j, -1.5)
        ref, ridx = O.nms(sb, 0.3, return_index=True)
        assert nk[j - 1] == ref.shape[0]
        assert np.array_equal(keep[j - 1, :nk[j - 1]], ref)
        assert np.array_equal(idx[j - 1, :nk[j - 1]], src[ridx])
        per.append(ref)
    kept, thr = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k, np.full((k.shape[0], 1), j + 1, np.float32)], 1) for j, k in enumerate(kept) if k.size])
    assert dets.shape == exp.shape and np.array_equal(dets, exp)


def test_host_mirror_tester(O, dev, small):
    """Tester_FRCNN.testOne / ImageDetect.detect orchestration (Python mirror) == fused device path."""
    from multipathnet_amd import detect
    net = small["net"]
    im, boxes = torch.from_numpy(small["im"]), torch.from_numpy(small["boxes"])
    tester = detect.Tester_FRCNN(net, opt={"test_nms_threshold": 0.3})
    img_boxes, (output, bbox_pred) = tester.testOne(im, boxes)
    net.test_one_async(im.to(dev), boxes.to(dev))
    keep, _, nk = [t.cpu() for t in net.nms_results()]
    for j, kb in enumerate(img_boxes):
        assert torch.equal(kb.cpu(), keep[j, : int(nk[j])])
    assert set(tester.last_timing) == {"forward", "nms", "total"}
    # iterative localisation (Tester_FRCNN.lua:82-89): second pass doubles the scored rows
    t2 = detect.Tester_FRCNN(net, opt={"test_num_iterative_loc": 2})
    _, (out2, bb2) = t2.testOne(im, boxes)
    assert out2.shape[0] == 2 * boxes.shape[0] and torch.equal(out2[: boxes.shape[0]], output)
    # the second pass ran on cached trunk features (recompute_features=false, ImageDetect.lua:107-111) and must equal
    # a full forward on the refined boxes
    new_boxes = t2.boxselect.forward([output, bbox_pred])
    full, _ = net.detect(im.to(dev), new_boxes, recompute_features=True)
    assert torch.equal(out2[boxes.shape[0]:], full)


def test_detect_is_deterministic_and_roi_order_equivariant(dev, small):
    net = small["net"]
    im, boxes = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    a, b = net.detect(im, boxes)
    a2, b2 = net.detect(im, boxes)
    assert torch.equal(a, a2) and torch.equal(b, b2)
    perm = torch.randperm(boxes.size(0), generator=torch.Generator().manual_seed(0)).to(dev)
    ap, bp = net.detect(im, boxes[perm].contiguous())
    assert torch.equal(ap, a[perm]) and torch.equal(bp, b[perm])   # rows are independent: exact
    sub, _ = net.detect(im, boxes[:77].contiguous())                  # memoryEfficientForward property (ImageDetect.lua:126-133)
    assert torch.equal(sub, a[:77])


@pytest.mark.parametrize("H,W", [(75, 125), (149, 251), (64, 96)])
def test_other_image_sizes_rezero_halo(O, dev, H, W):
    """changing the image size moves the zero halo of the C8P buffers; results must not depend on history"""
    from multipathnet_amd import models
    cfg = [8, 8, "P", 16, "P", 16]
    P = models.synthetic_params(cfg, pooled=7, fc_dim=32, n_classes=4, seed=1)
    net = models.FastRCNN(P, cfg=cfg, pooled=7, spatial_scale=0.25, max_h=150, max_w=256, max_rois=64)
    rng = np.random.default_rng(H)
    big = torch.from_numpy(rng.random((3, 150, 256), dtype=np.float32)).to(dev)
    net.detect(big, torch.from_numpy(_boxes(rng, 64, 256, 150)).to(dev))  # dirty the buffers at max size
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, 50, W, H)
    scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    Pn = _np_params(P)
    feat = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg)
    logits, deltas = O.frcnn_head(feat, O.project_im_rois(boxes, 1.0), Pn, pooled=7, spatial_scale=0.25)
    assert np.abs(scores.cpu().numpy() - O.softmax(logits)).max() < 1e-4
    ref_bbox = O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)
    assert np.abs(bbox.cpu().numpy() - ref_bbox).max() < 1e-4 * W


def test_ragged_shapes_one_handle(O, dev):
    """one handle, a sequence of odd image sizes and ROI counts (1 ROI, counts that are not a multiple of the 4-ROI pooling
    groups, maps narrower than one conv tile, a 1-tile first layer): scores, boxes and the whole testOne tail vs the oracle"""
    from multipathnet_amd import models
    cfg = [8, 8, "P", 16, "P", 16]
    P = models.synthetic_params(cfg, pooled=7, fc_dim=32, n_classes=5, seed=3)
    Pn = _np_params(P)
    net = models.FastRCNN(P, cfg=cfg, pooled=7, spatial_scale=0.25, max_h=160, max_w=256, max_rois=70)
    rng = np.random.default_rng(99)
    for (H, W, N) in [(160, 256, 70), (17, 23, 1), (33, 31, 3), (8, 32, 5), (9, 33, 6), (101, 7, 2), (64, 200, 69), (40, 40, 13)]:
        im = rng.random((3, H, W), dtype=np.float32)
        boxes = _boxes(rng, N, W, H, lo=2)
        imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
        scores, bbox = net.detect(imd, bd)
        feat = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg)
        logits, deltas = O.frcnn_head(feat, O.project_im_rois(boxes, 1.0), Pn, pooled=7, spatial_scale=0.25)
        assert scores.shape == (N, 5)
        assert np.abs(scores.cpu().numpy() - O.softmax(logits)).max() < 1e-4, (H, W, N)
        assert np.abs(bbox.cpu().numpy() - O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)).max() < 1e-4 * max(W, H), (H, W, N)
        dets, n = net.test_one_async(imd, bd)
        torch.cuda.synchronize()
        s, b = scores.cpu().numpy(), bbox.cpu().numpy()
        per = [O.nms(O.select_scored(s, b, j, -1.5)[0], 0.3) for j in range(1, 5)]
        kept, _ = O.keep_top_k(per, 100)
        rows = [np.concatenate([k, np.full((k.shape[0], 1), j + 1, np.float32)], 1) for j, k in enumerate(kept) if k.size]
        exp = np.concatenate(rows) if rows else np.zeros((0, 6), np.float32)
        assert np.array_equal(dets[: int(n.item())].cpu().numpy(), exp), (H, W, N)


def test_pipelined_equals_serial(dev, small):
    """mpn_frcnn_test_one_pipelined (NMS tail on the side stream, overlapping the next image) returns exactly what the
    serial form returns, image after image, with results valid one call later / after flush."""
    net = small["net"]
    rng = np.random.default_rng(99)
    ims = [torch.from_numpy(rng.random((3, SMALL["H"], SMALL["W"]), dtype=np.float32)).to(dev) for _ in range(5)]
    bxs = [torch.from_numpy(_boxes(rng, SMALL["N"] - 7 * i, SMALL["W"], SMALL["H"])).to(dev) for i in range(5)]
    serial = []
    for im, bx in zip(ims, bxs):
        d, n = net.test_one_async(im, bx)
        torch.cuda.synchronize()
        serial.append(d[: int(n.item())].clone())
    got, pending = [], None
    for im, bx in zip(ims, bxs):
        cur = net.test_one_pipelined(im, bx)
        if pending is not None:  # previous call's buffers are ordered on the stream now
            d, n = pending
            torch.cuda.current_stream().synchronize()
            got.append(d[: int(n.item())].clone())
        pending = cur
    net.flush()
    torch.cuda.current_stream().synchronize()
    d, n = pending
    got.append(d[: int(n.item())].clone())
    assert len(got) == len(serial)
    for a, b in zip(got, serial):
        assert a.shape == b.shape and torch.equal(a, b)
    # the serial entry point stays correct after pipelined use
    d, n = net.test_one_async(ims[0], bxs[0])
    torch.cuda.synchronize()
    assert torch.equal(d[: int(n.item())], serial[0])


def _np_tree(v):
    if isinstance(v, dict):
        return {k: _np_tree(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_np_tree(x) for x in v]
    return v.numpy() if hasattr(v, "numpy") else v


@pytest.mark.parametrize("conv345_norm", [True, False])
def test_multipathnet_head_vs_oracle(O, dev, conv345_norm):
    """BASELINE configs[2] graph at an oracle-sized scale: Foveal -> 5 towers (conv345Combine skip pooling + per-map L2
    normalise + 1x1 mix, fc6, fc7) -> K integral classifiers (mean of softmaxes) + het-tower box regressor
    (multipathnet.lua:64-120, model_utils.lua:209-251,275-317).  conv345_norm = False is opt.model_conv345_norm = false:
    MulConstant(1), (1/30), (1/200) per map instead of the L2 normalisation, no x1000 (model_utils.lua:222-223,239-241)."""
    from multipathnet_amd import models
    cfg = [8, 16, "P", 16, 24, "P", 32, 32, "P", 64, "P", 64]
    H, W, N, Cn, K = 150, 250, 120, 9, 3
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=11)
    P["conv345_norm"] = conv345_norm
    if not conv345_norm:  # unnormalised features are ~1000x smaller than the x1000 normalised ones: keep the head's inputs O(1)
        for T in P["towers"]:
            T["mix_w"] = T["mix_w"] * 300.0
    rng = np.random.default_rng(21)
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, N, W, H, lo=12)
    net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
    scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    Pn = _np_tree(P)
    taps = {}
    O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg, taps=taps)
    ref_scores, deltas = O.mpnet_head([taps["conv5"], taps["conv4"], taps["conv3"]], O.project_im_rois(boxes, 1.0), Pn)
    ref_bbox = O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)
    assert np.abs(scores.cpu().numpy().sum(1) - 1).max() < 1e-5
    assert np.abs(scores.cpu().numpy() - ref_scores).max() < 1e-4
    assert np.abs(bbox.cpu().numpy() - ref_bbox).max() < 1e-4 * W
    # the whole tail (select -> NMS -> top-k) runs on MultiPathNet outputs too
    dets, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    torch.cuda.synchronize()
    assert 0 < int(n.item()) <= dets.shape[0]


def test_multipathnet_pixel_major_pooling_equals_c8p_form(dev):
    """Round 3: the MultiPathNet head pools from PIXEL-MAJOR range-max tables with nn.Normalize's sum of squares fused into the
    pooling launch.  The pooled values are bit-identical to the C8P range-max kernel's (tests/test_gpu_roipool.py); the per-ROI
    norm is summed in a different (fixed) order, so scores agree to fp32 rounding, and each form is deterministic."""
    from multipathnet_amd import models
    cfg = [8, 16, "P", 16, 24, "P", 32, 32, "P", 64, "P", 64]
    H, W, N, Cn, K = 150, 250, 120, 9, 3
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=11)
    rng = np.random.default_rng(21)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    boxes = torch.from_numpy(_boxes(rng, N, W, H, lo=12)).to(dev)
    outs = []
    for pm in (1, 0):
        with hooks(roi_pool_pm=pm):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
            s1, b1 = net.detect(im, boxes)
            s2, b2 = net.detect(im, boxes)
            torch.cuda.synchronize()
            assert torch.equal(s1, s2) and torch.equal(b1, b2)
            outs.append((s1.clone(), b1.clone()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-6 and float((outs[0][1] - outs[1][1]).abs().max()) < 1e-3


def test_multipathnet_pooling_stream_overlap_is_invisible(dev):
    """Round 3: tower t + 1's skip pooling runs on the handle's pooling stream under tower t's GEMMs (two operand buffers, event
    hand-offs).  Pure scheduling: scores / boxes / detections are bit-identical to the single-stream order, call after call —
    a missing wait between the pooling stream and the launch stream would show up here as a mismatch or as run-to-run noise."""
    from multipathnet_amd import models
    cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
    H, W, N, Cn, K = 120, 200, 100, 6, 2
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=5)
    rng = np.random.default_rng(9)
    ims = [torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev) for _ in range(3)]
    bxs = [torch.from_numpy(_boxes(rng, N - 7 * i, W, H, lo=12)).to(dev) for i in range(3)]
    res = {}
    for ov in (1, 0):
        with hooks(pool_overlap=ov):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N, num_iter=2)
            outs = []
            for rep in range(4):
                for im, bx in zip(ims, bxs):
                    s1, b1 = net.detect(im, bx)
                    d, n = net.test_one_async(im, bx)
                    torch.cuda.synchronize()
                    outs.append((s1.clone(), b1.clone(), d[: int(n.item())].clone()))
            for rep in range(1, 4):
                for i in range(3):
                    assert all(torch.equal(x, y) for x, y in zip(outs[i], outs[3 * rep + i]))
            res[ov] = outs[:3]
    for a, b in zip(res[1], res[0]):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_multipathnet_mix_gemm_applies_the_normalisation(O, dev):
    """Round 3: where the mix GEMM runs un-split (always at BASELINE sizes; here a 384-channel conv5 makes it so at 100 ROIs), nn.Normalize's
    per-(ROI, map) scale is applied inside the GEMM — at the accumulator fold of each map's K segment (linear_c8_rowscaled) — instead of a
    read-modify-write pass over the pooled matrix.  Against the oracle (<= 1e-4), and against the in-place form (fp32 rounding)."""
    from multipathnet_amd import models
    cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
    H, W, N, Cn, K = 120, 200, 100, 6, 2
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=5)
    rng = np.random.default_rng(9)
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, N, W, H, lo=12)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    outs = []
    for fold in (1, 0):
        with hooks(mix_fold=fold):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
            s1, b1 = net.detect(imd, bd)
            s2, b2 = net.detect(imd, bd)
            torch.cuda.synchronize()
            assert torch.equal(s1, s2) and torch.equal(b1, b2)
            outs.append((s1.cpu().numpy(), b1.cpu().numpy()))
    assert 0 < np.abs(outs[0][0] - outs[1][0]).max() < 1e-5      # two different roundings of the same arithmetic — and really two code paths
    Pn = _np_tree(P)
    taps = {}
    O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg, taps=taps)
    ref_scores, deltas = O.mpnet_head([taps["conv5"], taps["conv4"], taps["conv3"]], O.project_im_rois(boxes, 1.0), Pn)
    ref_bbox = O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)
    for sc, bb in outs:
        assert np.abs(sc - ref_scores).max() < 1e-4 and np.abs(bb - ref_bbox).max() < 1e-4 * W


def test_alexnet_shaped_head_vs_oracle(O, dev):
    """BASELINE configs[0] head shape (models/alexnet.lua:23-27): ROIPooling(6,6,1/16) on a 256-channel map, fc6 9216->4096,
    300 selective-search-like ROIs, 21 classes, through the module-level C ABI, then clamp/select/NMS/top-k vs the oracle.
    (The AlexNet trunk itself — 11x11/4 and 5x5 grouped convs, LRN — is outside the 3x3 trunk this round builds.)"""
    from multipathnet_amd import nn, utils
    rng = np.random.default_rng(300)
    feat = np.maximum(rng.standard_normal((1, 256, 37, 62)), 0).astype(np.float32)
    boxes = _boxes(rng, 300, 1000, 600, lo=16, hi=500)
    rois = O.project_im_rois(boxes, 1.0)
    w6 = (rng.standard_normal((4096, 9216)) * (2.0 / 9216) ** 0.5).astype(np.float32); b6 = (rng.standard_normal(4096) * 0.01).astype(np.float32)
    w7 = (rng.standard_normal((4096, 4096)) * (2.0 / 4096) ** 0.5).astype(np.float32); b7 = (rng.standard_normal(4096) * 0.01).astype(np.float32)
    wc = (rng.standard_normal((21, 4096)) * 0.01).astype(np.float32); wb = (rng.standard_normal((84, 4096)) * 0.001).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pool = nn.ROIPooling(6, 6, 1 / 16)
    x = pool.forward([t(feat), t(rois)]).reshape(300, -1)
    def lin(x, w, b, relu):
        m = nn.Linear(w.shape[1], w.shape[0], relu=relu); m.weight = t(w); m.bias = t(b) if b is not None else None
        return m.forward(x)
    h7 = lin(lin(x, w6, b6, True), w7, b7, True)
    scores = nn.SoftMax().forward(lin(h7, wc, None, False))
    deltas = nn.BBoxNorm([0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2]).evaluate().forward(lin(h7, wb, None, False))
    dec = utils.decode_all_classes(t(boxes), deltas)
    # oracle
    px, _ = O.roi_pool(feat, rois, 6, 6, 1 / 16)
    r7 = O.linear(O.linear(px.reshape(300, -1), w6, b6, True), w7, b7, True)
    rs = O.softmax(O.linear(r7, wc, None))
    rd = O.bbox_decode(boxes, O.bbox_norm(O.linear(r7, wb, None), [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2]))
    assert np.array_equal(x.cpu().numpy(), px.reshape(300, -1))
    assert np.abs(scores.cpu().numpy() - rs).max() < 1e-4
    assert np.abs(dec.cpu().numpy() - rd).max() < 1e-4 * 1000


def test_fused_iterative_loc_and_bbox_voting(O, dev, small):
    """Tester_FRCNN.lua:82-99,118-124 fused on the device (num_iter = 2 on cached trunk features, per-class bbox voting)
    == the host mirror driving the module-level entry points == the oracle's nms/bbox_vote on the device's rows."""
    from multipathnet_amd import models, detect
    s = SMALL
    P = models.synthetic_params(s["cfg"], pooled=7, fc_dim=s["fc"], n_classes=s["C"], seed=557)
    net2 = models.FastRCNN(P, cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"],
                           num_iter=2, bbox_voting=True, bbox_vote_thresh=0.5)
    im, boxes = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    dets, n = net2.test_one_async(im, boxes)
    torch.cuda.synchronize()
    keep, idx, nk = [t.cpu().numpy() for t in net2.nms_results()]
    assert keep.shape[1] == 2 * s["N"]
    # host mirror on a plain pipeline with the same weights
    tester = detect.Tester_FRCNN(small["net"],
                                 opt={"test_num_iterative_loc": 2, "test_bbox_voting": True, "test_bbox_voting_nms_threshold": 0.5})
    img_boxes, (output, bbox_pred) = tester.testOne(im, boxes)
    assert output.shape[0] == 2 * s["N"]
    for j, kb in enumerate(img_boxes):
        assert np.array_equal(keep[j, : nk[j]], kb.cpu().numpy()), j
    # and against the oracle fed the device's rows
    sc, bb = output.cpu().numpy(), bbox_pred.cpu().numpy()
    per = []
    for j in range(1, s["C"]):
        sb, _ = O.select_scored(sc, bb, j, -1.5)
        ref = O.bbox_vote(O.nms(sb, 0.3), sb, 0.5)
        assert np.array_equal(keep[j - 1, : nk[j - 1]], ref, equal_nan=True)
        per.append(ref)
    kept, _ = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k, np.full((k.shape[0], 1), j + 1, np.float32)], 1) for j, k in enumerate(kept) if k.size])
    assert np.array_equal(dets[: int(n.item())].cpu().numpy(), exp)


@pytest.mark.parametrize("H0,W0", [(60, 100), (120, 90), (200, 320), (150, 250)])
def test_getimages_rescale_on_device(O, dev, H0, W0):
    """getImages (ImageDetect.lua:34-43) inside the pipeline: short side -> 150, long side capped at 250; ROIs projected with
    the scale, boxes decoded on the ORIGINAL boxes and clamped to the ORIGINAL image.  image.scale itself is parity-unpinned
    (external rock); the check is device == oracle restatement."""
    from multipathnet_amd import models
    cfg = [8, 8, "P", 16, "P", 16]
    P = models.synthetic_params(cfg, pooled=7, fc_dim=32, n_classes=4, seed=3)
    net = models.FastRCNN(P, cfg=cfg, pooled=7, spatial_scale=0.25, max_h=256, max_w=256, max_rois=40, scale=150, max_size=250)
    rng = np.random.default_rng(H0 * 7 + W0)
    im = rng.random((3, H0, W0), dtype=np.float32)
    boxes = _boxes(rng, 40, W0, H0, lo=6)
    scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    Pn = _np_params(P)
    s = O.pick_scale(H0, W0, 150, 250)
    x = O.image_transform(im, **O.ROSS)
    if s != 1.0:
        x = O.image_scale(x, int(H0 * s), int(W0 * s))
    feat = O.vgg_trunk(x, Pn["conv_w"], Pn["conv_b"], cfg)
    logits, deltas = O.frcnn_head(feat, O.project_im_rois(boxes, s), Pn, pooled=7, spatial_scale=0.25)
    ref_bbox = O.clamp_boxes(O.bbox_decode(boxes, deltas), W0, H0)
    assert np.abs(scores.cpu().numpy() - O.softmax(logits)).max() < 1e-4
    assert np.abs(bbox.cpu().numpy() - ref_bbox).max() < 1e-4 * max(W0, H0)


def test_image_scale_kernel_vs_oracle(O, dev):
    import ctypes as C
    from multipathnet_amd import nn, _lib
    rng = np.random.default_rng(4)
    for (h, w, h2, w2) in [(48, 64, 60, 80), (48, 64, 24, 32), (75, 125, 30, 100), (37, 53, 37, 90), (10, 10, 1, 1), (1, 7, 5, 7)]:
        im = rng.random((3, h, w), dtype=np.float32)
        d_in = torch.from_numpy(im).to(dev)
        tmp = torch.empty((3, h, w2), device=dev)
        out = torch.empty((3, h2, w2), device=dev)
        _lib.check(_lib.load().mpn_image_scale(nn._f(d_in), 3, h, w, h2, w2, nn._f(tmp), nn._f(out), None))
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy() - O.image_scale(im, h2, w2)).max() < 1e-6


def test_detect_unclamped_is_imagedetect_semantics(O, dev, small):
    """ImageDetect:detect returns UNCLAMPED decoded boxes (ImageDetect.lua:183-193); the clamp is Tester_FRCNN's (:75-78)."""
    from multipathnet_amd import detect
    s, net = SMALL, small["net"]
    im, boxes = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    sc_c, bb_c = net.detect(im, boxes)                    # clamp=True: what testOne holds after :75-78
    sc_u, bb_u = net.detect(im, boxes, clamp=False)
    assert torch.equal(sc_c, sc_u)
    ref_u = O.bbox_decode(small["boxes"], small["deltas"])
    assert np.abs(bb_u.cpu().numpy() - ref_u).max() < 1e-4 * s["W"]
    assert (np.abs(ref_u - O.clamp_boxes(ref_u, s["W"], s["H"])) > 0).any(), "the case must have boxes that leave the image"
    assert np.array_equal(O.clamp_boxes(bb_u.cpu().numpy(), s["W"], s["H"]), bb_c.cpu().numpy())
    d = detect.ImageDetect(net)
    _, bb_m = d.detect(im, boxes)
    assert torch.equal(bb_m, bb_u)
    with pytest.raises(ValueError):
        detect.ImageDetect(net, scale=[600], max_size=1000)   # the model's pipeline was built without getImages' rescaling


@pytest.mark.parametrize("num_iter,rbox,voting,score_pow", [(2, False, False, 1.0), (2, True, False, 1.0), (3, False, True, 1.0),
                                                            (3, False, True, 0.5), (3, False, True, 2.0), (1, False, True, 1.7),
                                                            (3, True, False, 1.0)])
def test_iterative_localisation_vs_oracle(O, dev, small, num_iter, rbox, voting, score_pow):
    """Tester_FRCNN.lua:72-100 against the ORACLE's restatement of the whole loop (first pass clamped, refinement passes not,
    SelectBoxes between passes, test_use_rbox_scores pairing): the device's joined score / box tables match it to the
    north_star tolerance, and the per-class NMS (+ voting) of the device's own rows is bit-exact — for the fused device path
    and for the host mirror.  opt.test_bbox_voting_score_pow (Tester_FRCNN.lua:119-121) != 1: the votes are weighted by
    score^p (double pow, rounded once — the oracle's restatement of scores:pow(p)), the kept boxes keep their NMS scores."""
    from multipathnet_amd import models, detect
    s = SMALL
    net = models.FastRCNN(small["P_torch"], cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"],
                          num_iter=num_iter, use_rbox_scores=rbox, bbox_voting=voting, bbox_vote_thresh=0.5, bbox_vote_score_pow=score_pow)
    im, boxes = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    dets, n = net.test_one_async(im, boxes)
    torch.cuda.synchronize()
    keep, idx, nk = [t.cpu().numpy() for t in net.nms_results()]
    rows = (num_iter - 1 if rbox else num_iter) * s["N"]
    assert keep.shape[1] == rows
    tester = detect.Tester_FRCNN(small["net"], opt={"test_num_iterative_loc": num_iter, "test_use_rbox_scores": rbox, "test_bbox_voting": voting,
                                                    "test_bbox_voting_nms_threshold": 0.5, "test_bbox_voting_score_pow": score_pow})
    img_boxes, (output, bbox_pred) = tester.testOne(im, boxes)
    assert output.shape[0] == rows and bbox_pred.shape[0] == rows
    for j, kb in enumerate(img_boxes):
        assert np.array_equal(keep[j, : nk[j]], kb.cpu().numpy(), equal_nan=True), j
    # the oracle's whole loop on the same inputs (its own fp32 summation order => tolerance on the tables)
    o_boxes, (osc, obb) = O.test_one(small["im"], small["boxes"], small["P"], num_iter=num_iter, use_rbox_scores=rbox, cfg=s["cfg"],
                                     target=s["H"], max_size=s["W"], bbox_voting=voting, bbox_vote_thresh=0.5, bbox_vote_score_pow=score_pow)
    sc, bb = output.cpu().numpy(), bbox_pred.cpu().numpy()
    assert np.abs(sc - osc).max() < 1e-4
    assert np.abs(bb - obb).max() < 2e-3 * s["W"]   # refinement passes decode from boxes that already carry the first pass's rounding
    n_first = s["N"]
    if not rbox and num_iter > 1:
        # the 2e-3 * W bound above is loose because each pass decodes from boxes that carry the previous pass's
        # rounding.  Pass by pass with the SAME inputs — the oracle is handed the boxes the device's own previous pass selected
        # (SelectBoxes of the device rows) — every refinement pass meets the single-pass tolerance.
        N = s["N"]
        for k in range(1, num_iter):
            new_boxes = O.select_boxes(sc[(k - 1) * N:k * N], bb[(k - 1) * N:k * N])
            s_o, dec_o, _, _ = O.detect(small["im"], new_boxes, small["P"], cfg=s["cfg"], target=s["H"], max_size=s["W"])
            assert np.abs(sc[k * N:(k + 1) * N] - s_o).max() < 1e-4, k
            assert np.abs(bb[k * N:(k + 1) * N] - dec_o).max() < 1e-4 * s["W"], k
    if not rbox and num_iter > 1:  # first-pass boxes are clamped, later passes are not
        assert bb[:n_first].min() >= 1.0 and bb[:n_first, 0::2].max() <= s["W"] and bb[:n_first, 1::2].max() <= s["H"]
        assert (bb[n_first:] < 1.0).any() or (bb[n_first:, 0::2] > s["W"]).any() or (bb[n_first:, 1::2] > s["H"]).any()
    # NMS / voting of the device's rows: bit-exact against the oracle (== compiled nms.c, tests/test_oracle_nms.py)
    per = []
    moved = False
    for j in range(1, s["C"]):
        sb, _ = O.select_scored(sc, bb, j, -1.5)
        ref = O.nms(sb, 0.3)
        if voting:
            votes = sb.copy()
            if score_pow != 1.0:
                votes[:, 4] = np.power(votes[:, 4].astype(np.float64), float(np.float32(score_pow))).astype(np.float32)
                assert not np.array_equal(votes[:, 4], sb[:, 4])
            ref = O.bbox_vote(ref, votes, 0.5)
            if score_pow != 1.0 and ref.shape[0]:  # the exponent really changes the vote (else the case tests nothing)
                moved = moved or not np.array_equal(ref, O.bbox_vote(O.nms(sb, 0.3), sb, 0.5))
        assert np.array_equal(keep[j - 1, : nk[j - 1]], ref, equal_nan=True)
        # the oracle's own loop end to end (its rows differ from the device's by summation order): same number of classes, and
        # where its rows select the same boxes the voted coordinates agree to the table tolerance
        assert len(o_boxes) == s["C"] - 1
        per.append(ref)
    if voting and score_pow != 1.0:
        assert moved
    kept, _ = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k, np.full((k.shape[0], 1), j + 1, np.float32)], 1) for j, k in enumerate(kept) if k.size])
    assert np.array_equal(dets[: int(n.item())].cpu().numpy(), exp, equal_nan=True)


def test_host_fed_pipeline_equals_device_fed(dev, small):
    """mpn_frcnn_test_one_pipelined_host (upload on the handle's copy stream into three staging sets) == the device-fed form,
    image after image, including a size change in the middle of the stream."""
    net = small["net"]
    rng = np.random.default_rng(7)
    s = SMALL
    ims = [rng.random((3, s["H"], s["W"]), dtype=np.float32) for _ in range(5)] + [rng.random((3, 120, 200), dtype=np.float32)]
    bxs = [_boxes(rng, s["N"], im.shape[2], im.shape[1]) for im in ims]
    ref = []
    for im, b in zip(ims, bxs):
        d, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(b).to(dev))
        torch.cuda.synchronize()
        ref.append(d[: int(n.item())].clone())
    pin = [(torch.from_numpy(im).pin_memory(), torch.from_numpy(b).pin_memory()) for im, b in zip(ims, bxs)]
    outs = [net.test_one_pipelined_host(i, b) for i, b in pin[:1]]
    got = []
    for i, b in pin[1:]:
        cur = net.test_one_pipelined_host(i, b)
        prev = outs[-1]
        got.append(prev[0][: int(prev[1].item())].clone())   # valid one call later (the .item() syncs the launch stream)
        outs.append(cur)
    net.flush()
    torch.cuda.synchronize()
    got.append(outs[-1][0][: int(outs[-1][1].item())].clone())
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_pipelined_host_fed_soak(dev, small):
    """240 images through the host-fed pipelined loop without a host sync in between (three different images and sizes in
    rotation, results read back only at the end from per-step output buffers): every step must reproduce the serial result of its
    image bit for bit — a race between the copy stream, the launch stream and the side stream's NMS tail shows up here"""
    net = small["net"]
    rng = np.random.default_rng(17)
    s = SMALL
    shapes = [(s["H"], s["W"], s["N"]), (120, 200, 150), (s["H"], s["W"], 64)]
    ims = [rng.random((3, h, w), dtype=np.float32) for h, w, _ in shapes]
    bxs = [_boxes(rng, n, w, h) for h, w, n in shapes]
    ref = []
    for im, b in zip(ims, bxs):
        d, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(b).to(dev))
        torch.cuda.synchronize()
        ref.append(d[: int(n.item())].clone())
    pin = [(torch.from_numpy(im).pin_memory(), torch.from_numpy(b).pin_memory()) for im, b in zip(ims, bxs)]
    steps = 240
    snap_d, snap_n = [], []
    prev = None
    for t_ in range(steps):
        cur = net.test_one_pipelined_host(*pin[t_ % 3])
        if prev is not None:  # image t-1's record is stream-ordered now: copy it aside on the launch stream (no host sync)
            snap_d.append(prev[0].clone()); snap_n.append(prev[1].clone())
        prev = cur
    net.flush()
    snap_d.append(prev[0].clone()); snap_n.append(prev[1].clone())
    torch.cuda.synchronize()
    assert len(snap_d) == steps
    for t_ in range(steps):
        n = int(snap_n[t_].item())
        assert n == ref[t_ % 3].shape[0], t_
        assert torch.equal(snap_d[t_][:n], ref[t_ % 3]), t_


def test_two_handles_two_threads_two_streams(dev, small):
    """SURVEY §8b 'Threading' / test_runner.lua:55-66: the reference drives one worker thread per GPU inside ONE process.
    No library state is process-global: two pipeline handles (different networks, so different split-K / NMS scratch sizes)
    driven concurrently from two host threads on two streams give exactly their serial results, repeatedly."""
    import threading
    from multipathnet_amd import models
    s = SMALL
    cfg_b = [8, 8, "P", 16, "P", 24, 24]
    Pb = models.synthetic_params(cfg_b, pooled=7, fc_dim=64, n_classes=11, seed=99)
    net_a = small["net"]
    net_b = models.FastRCNN(Pb, cfg=cfg_b, pooled=7, spatial_scale=0.25, max_h=100, max_w=180, max_rois=300)
    rng = np.random.default_rng(3)
    im_a, bx_a = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    im_b = torch.from_numpy(rng.random((3, 100, 180), dtype=np.float32)).to(dev)
    bx_b = torch.from_numpy(_boxes(rng, 300, 180, 100)).to(dev)

    def run(net, im, bx):
        d, n = net.test_one_async(im, bx)
        torch.cuda.current_stream().synchronize()
        return d[: int(n.item())].clone(), [t.clone() for t in net.detect(im, bx)]

    ref_a, ref_b = run(net_a, im_a, bx_a), run(net_b, im_b, bx_b)
    errs = []

    def worker(net, im, bx, ref):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(12):
                    d, (sc, bb) = run(net, im, bx)
                    if not (torch.equal(d, ref[0]) and torch.equal(sc, ref[1][0]) and torch.equal(bb, ref[1][1])):
                        errs.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(net_a, im_a, bx_a, ref_a)), threading.Thread(target=worker, args=(net_b, im_b, bx_b, ref_b))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs[:3]


def test_roi_pool_pixel_major_is_bit_identical(dev, small):
    """the pipeline pools from a pixel-major copy of the final map (one contiguous KiB per pixel and 256 channels); max is exact, so the
    pooled tensor and everything after it must equal the per-bin kernel's bit for bit — including degenerate and outside boxes"""
    from multipathnet_amd import models
    s = SMALL
    boxes = small["boxes"].copy()
    boxes[0] = [5, 5, 5, 5]                      # 1-pixel box
    boxes[1] = [s["W"] - 1, s["H"] - 1, s["W"], s["H"]]
    boxes[2] = [1, 1, s["W"], s["H"]]           # the whole image
    im, bx = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(boxes).to(dev)
    a_s, a_b = small["net"].detect(im, bx)
    a_p = small["net"].debug_tensor("pooled", small["pooled"].shape).clone()
    with hooks(roi_pool_pm=0):
        net0 = models.FastRCNN(small["P_torch"], cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"])
        b_s, b_b = net0.detect(im, bx)
        b_p = net0.debug_tensor("pooled", small["pooled"].shape).clone()
    assert torch.equal(a_p, b_p) and torch.equal(a_s, b_s) and torch.equal(a_b, b_b)


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [0, 2], ids=["launch-stream-heads", "side-stream-held-back"])
def test_deferred_heads_hazards(dev, small, defer):
    """Round 5: the pipelined forms of the plain Fast R-CNN head hand cls / bbox GEMM + softmax + decode + select over to the side stream
    after fc7 (they then run under the next image's first trunk layers).  What they read lives per buffer set (fc7's output, a copy of the
    caller's boxes, a split-K scratch of their own).  defer = 2 holds the side stream back 1 ms before every image's heads, so that the
    launch stream is a whole image ahead — its next fc6 / fc7 (split-K at this size), its next boxes upload, its next ROI projection all
    run BEFORE the previous image's heads: a missing guard shows as a wrong record.  defer = 0 is the previous form.  60 host-fed images of
    three sizes, every record against the serial form's, bit for bit."""
    from multipathnet_amd import models
    rng = np.random.default_rng(23)
    s = SMALL
    shapes = [(s["H"], s["W"], s["N"]), (120, 200, 150), (s["H"], s["W"], 64)]
    ims = [rng.random((3, h, w), dtype=np.float32) for h, w, _ in shapes]
    bxs = [_boxes(rng, n, w, h) for h, w, n in shapes]
    ref = []
    for im, b in zip(ims, bxs):   # the serial form on the product library
        d, n = small["net"].test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(b).to(dev))
        torch.cuda.synchronize()
        ref.append(d[: int(n.item())].clone())
    pin = [(torch.from_numpy(im).pin_memory(), torch.from_numpy(b).pin_memory()) for im, b in zip(ims, bxs)]
    with hooks(defer_heads=defer):   # the knob lives in the debug flavour: a handle of its own
        net = models.FastRCNN(small["P_torch"], cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"])
        steps, snap_d, snap_n, prev = 60, [], [], None
        for t_ in range(steps):
            cur = net.test_one_pipelined_host(*pin[t_ % 3])
            if prev is not None:
                snap_d.append(prev[0].clone()); snap_n.append(prev[1].clone())
            prev = cur
        net.flush()
        snap_d.append(prev[0].clone()); snap_n.append(prev[1].clone())
        torch.cuda.synchronize()
        # the un-pipelined entries stay correct straight after pipelined use (they join the side stream first)
        d, n = net.test_one_async(torch.from_numpy(ims[1]).to(dev), torch.from_numpy(bxs[1]).to(dev))
        torch.cuda.synchronize()
        assert torch.equal(d[: int(n.item())], ref[1])
    for t_ in range(steps):
        n = int(snap_n[t_].item())
        assert n == ref[t_ % 3].shape[0], t_
        assert torch.equal(snap_d[t_][:n], ref[t_ % 3]), t_


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["vggmpn", "rn_mpn_f32", "rn_mpn_bf16", "inc_mpn_bf16"])
def test_two_tower_lanes_are_invisible(dev, model):
    """Round 6: the towers of one image run on two LANES (towers 1, 3 on the handle's tower stream with their own activation buffers beside
    towers 0, 2, 4 on the caller's stream; the reference ran them on different GPUs, ModelParallelTable.lua:195-242).  Pure scheduling:
    scores, boxes and the detection record are bit-identical to the one-lane order (hook tower_lanes = 0), call after call, with different
    proposal counts in a row (a missing event between the lanes, the pooling stream and the launch stream would show up as a mismatch or
    as run-to-run noise), through detect(), the fused test_one and the pipelined form."""
    from multipathnet_amd import models
    rng = np.random.default_rng(5)
    if model == "vggmpn":
        cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
        H, W, N = 150, 250, 300
        P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=256, n_classes=9, n_integral=3, seed=11)
        mk = lambda: models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
    elif model.startswith("rn"):
        H, W, N = 150, 250, 200
        R = models.synthetic_resnet_mpn_params(depth=0, n_classes=7, n_integral=3, base_width=16, blocks=[1, 1, 1, 2], block_type="bottleneck", seed=31)
        mk = lambda: models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=N, top_k=20, bf16=model.endswith("bf16"))
    else:
        H, W, N = 170, 215, 120
        G = models.synthetic_inception_mpn_params(n_classes=5, n_integral=2, width=0.25, seed=17)
        mk = lambda: models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=N, top_k=10, bf16=True)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    bx = torch.from_numpy(_boxes(rng, N, W, H, lo=12)).to(dev)
    res = {}
    # (lanes, share): share = the "het" tower and the tower of the same Foveal region pool ONE operand (mpn_frcnn::tx3) — also pure scheduling /
    # buffer planning: the narrower tower's mix GEMM reads the K prefix of the wider one's pooled matrix, the same values it would pool itself
    # order = the towers run cheapest pooling first instead of in index order (each writes its own slice of the concat: the order is free)
    variants = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1), (0, 0, 1)] if model == "vggmpn" else [(0, 1, 1), (1, 1, 1)]
    for lanes, share, order in variants:
        # the graph towers' lanes are a debug-flavour experiment (tower_lanes = 2: measured without a gain, profiles/r06_tower_lanes_ab.txt);
        # the VGG MultiPathNet's are what ships (1)
        with hooks(tower_lanes=lanes * (1 if model == "vggmpn" else 2), tower_share=share, tower_order=order):
            net = mk()
            out = []
            for n in (N, N // 3, N, 7, N):
                s, b = net.detect(im, bx[:n].contiguous())
                out.append((s.clone(), b.clone()))
            for _ in range(3):
                dets, nd = net.test_one_async(im, bx)
                torch.cuda.synchronize()
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            bufs = [net.test_one_pipelined(im, bx) for _ in range(4)]
            net.flush()
            torch.cuda.synchronize()
            for dets, nd in bufs[-2:]:       # the two output sets alternate: the last two calls' records are both still there
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            res[(lanes, share, order)] = out
            del net
    base = res[variants[0]]
    for v in variants[1:]:
        assert len(res[v]) == len(base)
        for a, b in zip(base, res[v]):
            for x, y in zip(a, b):
                assert torch.equal(x, y), v
    last = res[variants[-1]]
    assert torch.equal(last[0][0], last[2][0]) and torch.equal(last[0][0], last[4][0])   # the same call gives the same rows every time


@pytest.mark.gpu
@pytest.mark.parametrize("rsi", [1, 0])
def test_packed_mix_rows_are_invisible(dev, rsi):
    """Round 6: the MultiPathNet mix GEMM's rows — (bin, roi) pairs — are packed, N rounded up to 8 per bin instead of to the fc operands' 128, and
    its epilogue scatters them into fc6's [cout block][bin][Mp][8] operand (1000 proposals: 383 row tiles instead of 392).  A row's sum does
    not depend on the tile it sits in: scores, boxes and detections are bit-identical to the padded layout (hook mix_packed = 0) for ROI counts
    that end inside a row tile, inside a group of 8, on a multiple of 128 and within 8 of one (where the two layouts coincide), in a row on one
    handle (the operand's pitch changes with the count), on the cached maps, with the in-place and the running-total row scales (gemm_rsi)."""
    from multipathnet_amd import models
    rng = np.random.default_rng(8)
    cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
    H, W, N = 150, 250, 300
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=256, n_classes=9, n_integral=3, seed=13)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    bx = torch.from_numpy(_boxes(rng, N, W, H, lo=6)).to(dev)
    res = {}
    for packed, share, lanes in [(0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0)]:
        with hooks(mix_packed=packed, tower_share=share, tower_lanes=lanes, gemm_rsi=rsi):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
            out = []
            for n in (N, 100, 7, 128, 125, 297, 1, 256, 250):
                s, b = net.detect(im, bx[:n].contiguous())
                out.append((s.clone(), b.clone()))
            s, b = net.detect(im, bx[40:141].contiguous(), recompute_features=False)
            out.append((s.clone(), b.clone()))
            for _ in range(2):
                dets, nd = net.test_one_async(im, bx)
                torch.cuda.synchronize()
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            bufs = [net.test_one_pipelined(im, bx[: (N if i % 2 else 203)].contiguous()) for i in range(4)]
            net.flush()
            torch.cuda.synchronize()
            for dets, nd in bufs[-2:]:
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            res[(packed, share, lanes)] = out
            del net
    base = res[(0, 1, 1)]
    for v, r in res.items():
        assert len(r) == len(base)
        for i, (a, b) in enumerate(zip(base, r)):
            for x, y in zip(a, b):
                assert torch.equal(x, y), (v, i)
    assert torch.equal(base[1][0], base[0][0][:100])          # and rows do not depend on the batch they are scored in


@pytest.mark.gpu
def test_range_max_tables_built_on_the_pooling_stream_are_invisible(dev):
    """Round 6: a map's range-max tables are built on the pooling stream where its first pooling is enqueued (conv5's in front of the first
    tower, conv4's / conv3's under that tower's GEMMs) instead of all of them on the launch stream in front of the head.  Pure scheduling:
    scores, boxes and detections are bit-identical to the up-front order (hook tables_lazy = 0), with and without the shared operand, with a
    three-map tower first (tower_order = 0), for ROI counts in a row and on the cached maps + tables of the iterative-localisation path."""
    from multipathnet_amd import models
    rng = np.random.default_rng(6)
    cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
    H, W, N = 180, 290, 300
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=256, n_classes=9, n_integral=3, seed=12)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    bx = torch.from_numpy(_boxes(rng, N, W, H, lo=4)).to(dev)
    res = {}
    for lazy, share, order in [(0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (1, 0, 0)]:
        with hooks(tables_lazy=lazy, tower_share=share, tower_order=order):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
            out = []
            for n in (N, 40, N // 2, 3):
                s, b = net.detect(im, bx[:n].contiguous())
                out.append((s.clone(), b.clone()))
                s, b = net.detect(im, bx[n // 2:n // 2 + 50].contiguous(), recompute_features=False)   # the cached maps AND their tables (ImageDetect.lua:107-111)
                out.append((s.clone(), b.clone()))
            for _ in range(2):
                dets, nd = net.test_one_async(im, bx)
                torch.cuda.synchronize()
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            bufs = [net.test_one_pipelined(im, bx) for _ in range(4)]
            net.flush()
            torch.cuda.synchronize()
            for dets, nd in bufs[-2:]:
                out.append((dets[: int(nd.item())].clone(), nd.clone()))
            res[(lazy, share, order)] = out
            del net
    base = res[(0, 1, 1)]
    for v, r in res.items():
        assert len(r) == len(base)
        for a, b in zip(base, r):
            for x, y in zip(a, b):
                assert torch.equal(x, y), v


@pytest.mark.gpu
def test_fc6_three_plane_split_vs_oracle_and_fp32_pipeline(O, dev, small):
    """MPN_FC_SPLIT3 (VERDICT r5 task 2; models/vgg.lua:16,30): fc6 on the bf16 matrix pipe — both operands split exactly into three bf16 planes,
    the six plane products of weight >= 2^-16 accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Held to the gate of the fp32 path: logits and
    deltas within 1e-4 of the oracle, scores within 1e-4, and no further from a float64 head than the fp32-MFMA pipeline is (x 1.2).  Rows do
    not depend on the batch they are scored in (fixed K ranges)."""
    from multipathnet_amd import models
    s = SMALL
    net = models.FastRCNN(small["P_torch"], cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"], fc_arith="split3")
    assert net.fc_arith == 1
    im, bx = torch.from_numpy(small["im"]).to(dev), torch.from_numpy(small["boxes"]).to(dev)
    scores, bbox = net.detect(im, bx)
    torch.cuda.synchronize()
    cls = net.debug_tensor("cls", small["logits"].shape).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", small["deltas"].shape).cpu().numpy()
    assert np.abs(cls - small["logits"]).max() < 1e-4 and np.abs(raw - small["deltas"]).max() < 1e-4
    assert np.abs(scores.cpu().numpy() - O.softmax(small["logits"])).max() < 1e-4
    # against a float64 head on the device's own pooled operand: the split's error beside the fp32 MFMA path's
    ref = small["net"]
    ref.detect(im, bx)
    torch.cuda.synchronize()
    pooled = net.debug_tensor("pooled", small["pooled"].shape).cpu().numpy().reshape(s["N"], -1).astype(np.float64)
    P = small["P"]
    h6 = np.maximum(pooled @ P["fc6_w"].astype(np.float64).T + P["fc6_b"], 0)
    h7 = np.maximum(h6 @ P["fc7_w"].astype(np.float64).T + P["fc7_b"], 0)
    l64 = h7 @ P["cls_w"].astype(np.float64).T + P["cls_b"]
    e_split = np.abs(cls - l64).max()
    e_fp32 = np.abs(ref.debug_tensor("cls", small["logits"].shape).cpu().numpy() - l64).max()
    print("logits vs a float64 head on the same pooled operand: three-plane split %.3g, fp32 MFMA %.3g" % (e_split, e_fp32))
    assert e_split < max(1.2 * e_fp32, 1e-6)
    # batch invariance + determinism
    s2, b2 = net.detect(im, bx[:37].contiguous(), recompute_features=False)
    assert torch.equal(s2, scores[:37]) and torch.equal(b2, bbox[:37])
    s3, b3 = net.detect(im, bx)
    assert torch.equal(s3, scores) and torch.equal(b3, bbox)


@pytest.mark.gpu
def test_mix_gemm_row_scales_in_place_equal_the_running_total_form(O, dev):
    """Round 6: MultiPathNet's mix GEMM applies nn.Normalize's per-(map, ROI) scales by moving the accumulator from one K segment's scale to the
    next (gemm_c8_pf_kernel<4, false, true>: 64 fewer registers than the running-total form, so its blocks share a CU with the other tower lane's
    fc6) instead of folding scaled segments into a running total (hook gemm_rsi = 0: rounds 3-5).  Same sum with two or three more roundings per
    element: scores agree to fp32 rounding, both forms are deterministic, and both are within the path's 1e-4 of the oracle."""
    from multipathnet_amd import models
    cfg = [16, 32, "P", 32, 64, "P", 64, 96, "P", 128, "P", 384]
    H, W, N, Cn, K = 150, 250, 150, 9, 3
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=256, n_classes=Cn, n_integral=K, seed=11)
    rng = np.random.default_rng(21)
    im_np = rng.random((3, H, W), dtype=np.float32)
    bx_np = _boxes(rng, N, W, H, lo=12)
    im, bx = torch.from_numpy(im_np).to(dev), torch.from_numpy(bx_np).to(dev)
    outs = []
    for rsi in (1, 0):
        with hooks(gemm_rsi=rsi):
            net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
            s1, b1 = net.detect(im, bx)
            s2, b2 = net.detect(im, bx)
            torch.cuda.synchronize()
            assert torch.equal(s1, s2) and torch.equal(b1, b2)
            outs.append((s1.clone(), b1.clone()))
            del net
    d = float((outs[0][0] - outs[1][0]).abs().max())
    print("scores, in-place row scales vs running total: max |d| = %.3g" % d)
    assert d < 2e-6 and float((outs[0][1] - outs[1][1]).abs().max()) < 1e-3
    Pn = _np_tree(P)
    taps = {}
    O.vgg_trunk(O.image_transform(im_np, **O.ROSS), Pn["conv_w"], Pn["conv_b"], cfg, taps=taps)
    ref_scores, _ = O.mpnet_head([taps["conv5"], taps["conv4"], taps["conv3"]], O.project_im_rois(bx_np, 1.0), Pn)
    assert np.abs(outs[0][0].cpu().numpy() - ref_scores).max() < 1e-4
