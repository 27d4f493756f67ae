"""oracle restatement of the ResNet layers (models/resnet.lua graph; the arithmetic lives in external nn / cudnn, source
absent => PARITY UNPINNED): cross-checked against PyTorch-CPU fp32, which implements the same public definitions."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize("B,ci,co,h,w,k,s,p", [(1, 3, 8, 23, 31, 7, 2, 3), (2, 8, 16, 14, 14, 1, 1, 0), (3, 8, 8, 14, 14, 3, 2, 1),
                                               (1, 16, 8, 9, 11, 3, 1, 1), (2, 8, 24, 7, 7, 1, 2, 0)])
def test_conv2d_vs_torch(O, B, ci, co, h, w, k, s, p):
    rng = np.random.default_rng(k * 100 + s * 10 + p)
    x = rng.standard_normal((B, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, k, k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(b), stride=s, padding=p).numpy()
    res = rng.standard_normal(ref.shape).astype(np.float32)
    y = O.conv2d(x, wt, b, stride=s, pad=p, relu=False)
    assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-4
    y2 = O.conv2d(x, wt, b, stride=s, pad=p, relu=True, residual=res)
    assert np.abs(y2 - np.maximum(ref + res, 0)).max() < 1e-4


def test_pools_vs_torch(O):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 37, 50)).astype(np.float32)
    assert np.array_equal(O.maxpool2d(x, 3, 2, 1), F.max_pool2d(torch.from_numpy(x), 3, 2, 1).numpy())
    z = rng.standard_normal((4, 6, 7, 7)).astype(np.float32)
    assert np.abs(O.avgpool_global(z) - z.mean((2, 3))).max() < 1e-6


@pytest.mark.parametrize("bt", ["basic", "bottleneck"])
def test_resnet_graph_vs_torch(O, bt):
    """the composed trunk / head (block order, where the stride and the shortcut sit, ReLU placement) against a direct
    PyTorch transcription of fb.resnet.torch's basicblock / bottleneck"""
    from multipathnet_amd import models
    R = models.synthetic_resnet_params(depth=0, n_classes=4, base_width=8, blocks=[1, 2, 1, 1], block_type=bt, seed=11)
    Rn = models.resnet_params_numpy(R)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 64, 80)).astype(np.float32)

    def block(t, b):
        sc = t if b["shortcut"] is None else F.conv2d(t, b["shortcut"][0], b["shortcut"][1], stride=b["shortcut"][2])
        y = t
        for i, (w, bb, st, pd) in enumerate(b["convs"]):
            y = F.conv2d(y, w, bb, stride=st, padding=pd)
            y = F.relu(y + sc) if i == len(b["convs"]) - 1 else F.relu(y)
        return y

    t = F.relu(F.conv2d(torch.from_numpy(x)[None], R["conv1_w"], R["conv1_b"], stride=2, padding=3))
    t = F.max_pool2d(t, 3, 2, 1)
    for b in R["trunk_blocks"]:
        t = block(t, b)
    feat = O.resnet_trunk(x, Rn)
    assert feat.shape == tuple(t.shape[1:]) and np.abs(feat - t[0].numpy()).max() < 1e-4 * max(1.0, float(t.abs().max()))
    rois = np.array([[1, 1, 1, 60, 50], [1, 10, 5, 79, 63], [1, 30, 30, 31, 31]], np.float32)
    logits, deltas = O.resnet_head(feat, rois, Rn, pooled=6)
    pooled, _ = O.roi_pool(feat, rois, 6, 6, 1.0 / 16)
    h = torch.from_numpy(pooled)
    for b in R["head_blocks"]:
        h = block(h, b)
    f = h.mean((2, 3))
    assert np.abs(logits - (f @ R["cls_w"].T + R["cls_b"]).numpy()).max() < 1e-5


def test_rect_conv_and_avgpool_vs_torch(O):
    """asymmetric kernels (1x7 / 7x1 / 1x3 / 3x1) and count_include_pad average pooling of the Inception graph vs PyTorch-CPU"""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 8, 17, 13)).astype(np.float32)
    for (kh, kw, sh, sw, ph, pw) in [(1, 7, 1, 1, 0, 3), (7, 1, 1, 1, 3, 0), (1, 3, 1, 1, 0, 1), (3, 1, 1, 1, 1, 0), (3, 3, 2, 2, 0, 0), (5, 5, 1, 1, 2, 2)]:
        w = (rng.standard_normal((16, 8, kh, kw)) * 0.1).astype(np.float32)
        b = rng.standard_normal(16).astype(np.float32)
        ref = F.relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=(sh, sw), padding=(ph, pw))).numpy()
        assert np.abs(O.conv2d_rect(x, w, b, sh, sw, ph, pw, True) - ref).max() < 1e-4
    ref = F.avg_pool2d(torch.from_numpy(x), 3, 1, 1, count_include_pad=True).numpy()
    assert np.abs(O.avgpool2d(x, 3, 1, 1) - ref).max() < 1e-6


def test_inception_graph_builder_shapes():
    """the op lists reproduce Inception-v3's channel arithmetic: 768-channel 17x17 stage, 2048-channel head, 299 -> 17 -> 8"""
    from multipathnet_amd import models
    G = models.synthetic_inception_v3_params(n_classes=3, width=1.0, seed=1)
    assert G["trunk_tensor_c"][G["feat_tensor"]] == 768 and G["head_tensor_c"][G["out_tensor"]] == 2048
    assert sum(1 for o in G["trunk_ops"] + G["head_ops"] if o["kind"] == 0) == 94  # the 94 convolutions of Inception-v3 (without the aux head)
    h = 299
    dims = {0: h}
    for o in G["trunk_ops"]:
        dims.setdefault(o["dst"], (dims[o["src"]] + 2 * o["ph"] - o["kh"]) // o["sh"] + 1)
    assert dims[G["feat_tensor"]] == 17
    dims = {0: 17}
    for o in G["head_ops"]:
        dims.setdefault(o["dst"], (dims[o["src"]] + 2 * o["ph"] - o["kh"]) // o["sh"] + 1)
    assert dims[G["out_tensor"]] == 8


def test_average_pool_commutes_with_pointwise_convolution(O):
    """the identity behind graph_parse's pool-branch rewrite (resnet.hip): a count_include_pad 3x3/1 average pool and a 1x1
    convolution are both linear and the pool pads with zeros, so relu(conv(pool(x)) + b) == relu(pool(conv(x)) + b) up to fp32
    summation order — including the border cells, where the window overlaps the padding"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 40, 8, 8)).astype(np.float32)
    w = (rng.standard_normal((24, 40, 1, 1)) * 0.2).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    ref = O.conv2d(O.avgpool2d(x, 3, 1, 1), w, b, relu=True)
    pooled = O.avgpool2d(O.conv2d(x, w, None), 3, 1, 1)
    alt = np.maximum(pooled + b[None, :, None, None], 0.0)
    assert np.abs(ref - alt).max() < 1e-5 * max(1.0, np.abs(ref).max())
    # with bf16 storage the two orders differ only by where the intermediate is rounded
    xb = O.bf16_round(x)
    refb = O.conv2d(O.bf16_round(O.avgpool2d(xb, 3, 1, 1)), w, b, relu=True, bf16=True)
    altb = O.bf16_round(np.maximum(O.avgpool2d(O.conv2d(xb, w, None, bf16=True), 3, 1, 1) + b[None, :, None, None], 0.0))
    assert np.abs(refb - altb).max() < 2e-2 * max(1.0, np.abs(refb).max())
