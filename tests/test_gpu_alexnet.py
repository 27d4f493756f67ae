"""BASELINE configs[0]: AlexNet / CaffeNet Fast R-CNN (models/alexnet.lua:14-27) through the op-list pipeline — grouped convolutions as
channel-range ops, cross-channel LRN, ceil-mode max-pooling, inn.ROIPooling(6,6,1/16), fc6 / fc7 as a 6x6 / 1x1 convolution —
against the oracle (cross-checked against PyTorch-CPU in tests/test_oracle_alexnet.py), at test size and at the config's full size
(600x1000 image, 300 ROIs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(H, W, N, seed):
    rng = np.random.default_rng(seed)
    im = rng.random((3, H, W), dtype=np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(12), np.log(min(H, W)), (N, 2)))
    boxes = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)
    return im, boxes


@pytest.mark.parametrize("H,W,N,width", [(150, 250, 40, 0.25), (131, 97, 25, 0.5)])
def test_alexnet_frcnn_vs_oracle(O, dev, H, W, N, width):
    from multipathnet_amd import models
    G = models.synthetic_alexnet_params(n_classes=6, width=width, fc_dim=128, seed=H)
    Gn = models.graph_params_numpy(G)
    im, boxes = _inputs(H, W, N, W)
    net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=64, top_k=20)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    so, bo, _, _ = O.graph_detect(im, boxes, Gn, O.ROSS, target=min(H, W), max_size=max(H, W), pooled=6, spatial_scale=1.0 / 16)
    assert np.abs(s.cpu().numpy() - so).max() < 1e-4
    assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < 1e-4 * max(H, W)
    # the whole testOne path runs on it too: per-class NMS of the device's rows == the oracle's
    dets, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    torch.cuda.synchronize()
    keep, _, nk = [t.cpu().numpy() for t in net.nms_results()]
    sn, bn = s.cpu().numpy(), b.cpu().numpy()
    for j in range(1, 6):
        sb, _ = O.select_scored(sn, bn, j, -1.5)
        ref = O.nms(sb, 0.3)
        assert nk[j - 1] == ref.shape[0] and np.array_equal(keep[j - 1, : nk[j - 1]], ref)


@pytest.mark.parametrize("fuse", [511, 255, 7])
def test_alexnet_cpu_branch_roi_pooling_vs_oracle(O, dev, fuse):
    """BASELINE configs[0] says "CPU nn path": inn.ROIPooling's CPU branch crops the clipped window and runs nn.SpatialAdaptiveMaxPooling
    (models/alexnet.lua:23 with float tensors).  roi_bin_rule = MPN_ROI_BINS_ADAPTIVE through the op-list pipeline — the pixel-major pooling
    into the fully-connected operand (511), the row-per-thread C8I kernel writing that operand (255), the C8I batch of maps feeding the
    fc-as-convolution form (7) — against the oracle's graph restatement pooling with the same rule; and it differs from the default rule."""
    from conftest import hooks
    from multipathnet_amd import models
    H, W, N = 150, 250, 40
    G = models.synthetic_alexnet_params(n_classes=6, width=0.25, fc_dim=128, seed=H)
    Gn = models.graph_params_numpy(G)
    im, boxes = _inputs(H, W, N, W)
    boxes[:8, 2:] = [W + 40, H + 40]   # proposals that hang over the right / bottom border (CaffeNet's conv5 map of a 150 x 250 image is 11 x 17: its
    boxes[8:12, :2] = [-30, -30]       # last column is 16 = round(249 / 16), so in-image boxes never leave it): the rounded window leaves the map
    outs = {}
    with hooks(graph_fuse=fuse):
        for rule in (1, 0):
            net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=64, top_k=20, roi_bin_rule=rule)
            s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
            with O.roi_bin_rule(rule):
                so, bo, _, _ = O.graph_detect(im, boxes, Gn, O.ROSS, target=min(H, W), max_size=max(H, W), pooled=6, spatial_scale=1.0 / 16)
            assert np.abs(s.cpu().numpy() - so).max() < 1e-4, (fuse, rule)
            assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < 1e-4 * max(H, W)
            outs[rule] = s.cpu().numpy().copy()
            del net
    assert np.abs(outs[0] - outs[1]).max() > 1e-5


def test_alexnet_fc_layers_on_the_gemm_equal_the_convolution_form(dev):
    """graph_parse bit 3: fc6 (the 6x6 convolution over the whole pooled map) and fc7 (1x1 on 1x1 maps) run on the tuned GEMM — the
    ROI pooling writes (bin, roi) rows, K = (channel block, bin) — instead of the pixel-tile convolution kernels; bit 4: conv1 (3 input
    channels) as a GEMM over im2col rows, k = (tap, channel), instead of 121 taps of a 3-of-8-channel record: the same dot products in
    another summation order; bit 5: conv3 / conv4 / conv5 (3x3, stride 1, pad 1; the grouped ones as channel ranges of the padded
    map) on the VGG pipeline's Winograd F(2x2,3x3) kernel"""
    from conftest import hooks
    from multipathnet_amd import models
    H, W, N = 150, 250, 40
    G = models.synthetic_alexnet_params(n_classes=6, width=0.25, fc_dim=128, seed=3)
    im, boxes = _inputs(H, W, N, 4)
    out = []
    for fuse in (7, 15, 31, 63, 63 + 256):
        with hooks(graph_fuse=fuse):
            net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=64, top_k=20)
            s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
            out.append((s.cpu().numpy().copy(), b.cpu().numpy().copy()))
            del net
    for a, b, tol in ((0, 1, 1e-6), (1, 2, 1e-6), (2, 3, 1e-5)):
        assert not np.array_equal(out[a][0], out[b][0])        # really two code paths
        assert np.abs(out[a][0] - out[b][0]).max() < tol and np.abs(out[a][1] - out[b][1]).max() < 1e3 * tol
    # bit 8: the (bin, roi)-row operand pooled by the VGG pipeline's pixel-major kernel instead of the row-per-thread one: the same max
    assert np.array_equal(out[4][0], out[3][0]) and np.array_equal(out[4][1], out[3][1])


def test_alexnet_smaller_image_after_a_larger_one(dev):
    """The Winograd convolutions read their zero padding from the halo of a padded copy of the map; a handle that has seen a larger
    image re-lays the halo for a smaller one (graph_run: fit_halo): same result as a fresh handle"""
    from multipathnet_amd import models
    G = models.synthetic_alexnet_params(n_classes=6, width=0.25, fc_dim=128, seed=8)
    big, bb = _inputs(150, 250, 30, 1)
    small, sb = _inputs(131, 97, 25, 2)
    net = models.AlexNetFRCNN(G, max_h=150, max_w=250, max_rois=64, top_k=20)
    net.detect(torch.from_numpy(big).to(dev), torch.from_numpy(bb).to(dev))
    s1, b1 = [t.clone() for t in net.detect(torch.from_numpy(small).to(dev), torch.from_numpy(sb).to(dev))]
    s3, b3 = [t.clone() for t in net.detect(torch.from_numpy(big).to(dev), torch.from_numpy(bb).to(dev))]
    fresh = models.AlexNetFRCNN(G, max_h=150, max_w=250, max_rois=64, top_k=20)
    s2, b2 = fresh.detect(torch.from_numpy(small).to(dev), torch.from_numpy(sb).to(dev))
    assert torch.equal(s1, s2) and torch.equal(b1, b2)
    s4, b4 = models.AlexNetFRCNN(G, max_h=150, max_w=250, max_rois=64, top_k=20).detect(torch.from_numpy(big).to(dev), torch.from_numpy(bb).to(dev))
    assert torch.equal(s3, s4) and torch.equal(b3, b4)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_alexnet_sharded_equals_unsharded(dev, world):
    """With fc6 / fc7 on the row-invariant GEMM the AlexNet head's rows no longer depend on the batch they are scored in: the
    ROI-sharded latency mode (tests/test_gpu_shard.py) reproduces the unsharded detections bit for bit on this graph pipeline too"""
    from multipathnet_amd import models
    from test_gpu_shard import _emulate, _reference
    H, W, N = 150, 250, 61
    G = models.synthetic_alexnet_params(n_classes=6, width=0.25, fc_dim=128, seed=5)
    im, boxes = _inputs(H, W, N, 6)
    net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=64, top_k=20)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    ref_dets, keep, kidx, nk = _reference(net, imd, bd)
    dets, n, _, _ = _emulate(net, imd, bd, world)
    assert n == ref_dets.size(0) and n > 0 and torch.equal(dets, ref_dets)


def test_alexnet_fullsize_config0(O, dev):
    """configs[0] at full size: 600x1000 image, 300 ROIs, 21 classes, full-width CaffeNet (conv5 map 39 x 64 at stride 16).  The oracle
    runs the whole trunk and the head of a 60-ROI sample; logits / deltas within 1e-4 absolute."""
    from multipathnet_amd import models
    H, W, N = 600, 1000, 300
    G = models.synthetic_alexnet_params(n_classes=21, seed=557)
    Gn = models.graph_params_numpy(G)
    im, boxes = _inputs(H, W, N, 556)
    net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=N)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s, b = s.cpu().numpy(), b.cpu().numpy()
    assert np.abs(s.sum(1) - 1).max() < 1e-5
    idx = np.random.default_rng(5).choice(N, 60, replace=False)
    so, bo, logits, deltas = O.graph_detect(im, boxes[idx], Gn, O.ROSS, pooled=6, spatial_scale=1.0 / 16)
    assert np.abs(s[idx] - so).max() < 1e-4
    assert np.abs(b[idx] - O.clamp_boxes(bo, W, H)).max() < 1e-2
    raw = net.debug_tensor("bbox_raw", (N, 4 * 21)).cpu().numpy()
    assert np.abs(raw[idx] - deltas).max() < 1e-4
    dets, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    torch.cuda.synchronize()
    assert 0 < int(n.item()) <= dets.size(0)
