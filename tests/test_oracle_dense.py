"""Cross-checks the 'parity unpinned' dense restatements (conv / ceil max-pool / linear / softmax /
normalize) against PyTorch-CPU, and the Lua-source-pinned geometry modules against independent
numpy restatements of the cited Lua lines."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_conv3x3_vs_torch(O):
    rng = np.random.default_rng(0)
    for (ci, co, h, w) in [(3, 8, 9, 13), (16, 32, 40, 56), (8, 8, 1, 1), (5, 7, 2, 33)]:
        x = rng.standard_normal((ci, h, w)).astype(np.float32)
        wt = (rng.standard_normal((co, ci, 3, 3)) * 0.2).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        y = O.conv3x3(x, wt, b, relu=True)
        yt = F.relu(F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(wt), torch.from_numpy(b), padding=1))[0].numpy()
        assert np.abs(y - yt).max() < 1e-4 * max(1.0, np.abs(yt).max())


def test_maxpool_ceil_vs_torch(O):
    rng = np.random.default_rng(1)
    for (c, h, w) in [(4, 75, 125), (3, 38, 63), (2, 1, 1), (2, 2, 3), (1, 600, 10)]:
        x = rng.standard_normal((c, h, w)).astype(np.float32)
        p = O.maxpool2x2_ceil(x)
        pt = F.max_pool2d(torch.from_numpy(x)[None], 2, 2, ceil_mode=True)[0].numpy()
        assert p.shape == pt.shape and np.array_equal(p, pt)


def test_vgg_feature_map_is_38_rows(O):
    # test.lua:142 — a 600-px side gives a 38-row conv5 map (4 ceil-mode pools): 600->300->150->75->38
    h = 600
    for _ in range(4):
        h = (h + 1) // 2
    assert h == 38
    x = np.zeros((1, 75, 125), np.float32)
    assert O.maxpool2x2_ceil(x).shape == (1, 38, 63)


def test_linear_softmax_vs_torch(O):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((37, 300)).astype(np.float32)
    w = (rng.standard_normal((70, 300)) * 0.1).astype(np.float32)
    b = rng.standard_normal(70).astype(np.float32)
    y = O.linear(x, w, b, relu=True)
    yt = F.relu(F.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))).numpy()
    assert np.abs(y - yt).max() < 1e-4
    s = O.softmax(y[:, :21])
    st = torch.softmax(torch.from_numpy(y[:, :21]), 1).numpy()
    assert np.abs(s - st).max() < 1e-6
    n = O.l2_normalize(x)
    nt = F.normalize(torch.from_numpy(x), dim=1).numpy()
    assert np.abs(n - nt).max() < 1e-6


def test_linear_chunk_invariance(O):
    # test.lua:165-179 — chunked (25) == un-chunked exactly
    rng = np.random.default_rng(4)
    x = rng.standard_normal((40, 512)).astype(np.float32)
    w = rng.standard_normal((9, 512)).astype(np.float32)
    b = rng.standard_normal(9).astype(np.float32)
    full = O.linear(x, w, b)
    parts = np.concatenate([O.linear(x[:25], w, b), O.linear(x[25:], w, b)])
    assert np.array_equal(full, parts)


def _roi_pool_numpy(feat, rois, PH, PW, scale):
    """independent restatement (vectorised differently) of the documented ROI-pool semantics"""
    N, (C, H, W) = rois.shape[0], feat.shape[1:]
    out = np.zeros((N, C, PH, PW), np.float32)
    f32 = np.float32
    for n in range(N):
        r = rois[n]
        rnd = lambda v: int(np.floor(abs(v) + f32(0.5)) * np.sign(v))  # roundf: half away from zero
        sw, sh = rnd(f32(f32(r[1] - f32(1)) * f32(scale))), rnd(f32(f32(r[2] - f32(1)) * f32(scale)))
        ew, eh = rnd(f32(f32(r[3] - f32(1)) * f32(scale))), rnd(f32(f32(r[4] - f32(1)) * f32(scale)))
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bw, bh = f32(rw) / f32(PW), f32(rh) / f32(PH)
        for ph in range(PH):
            for pw in range(PW):
                hs = int(np.floor(f32(ph) * bh)) + sh; he = int(np.ceil(f32(ph + 1) * bh)) + sh
                ws = int(np.floor(f32(pw) * bw)) + sw; we = int(np.ceil(f32(pw + 1) * bw)) + sw
                hs, he = min(max(hs, 0), H), min(max(he, 0), H)
                ws, we = min(max(ws, 0), W), min(max(we, 0), W)
                if he > hs and we > ws:
                    out[n, :, ph, pw] = feat[int(r[0]) - 1, :, hs:he, ws:we].max(axis=(1, 2))
    return out


def test_roi_pool_semantics(O):
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((2, 6, 38, 50)).astype(np.float32)
    # test.lua:141-146 style ROIs: randn*50 (negative / malformed boxes included), batch index valid
    rois = (rng.standard_normal((40, 5)) * 50).astype(np.float32)
    rois[:, 0] = rng.integers(1, 3, 40)
    good = np.array([[1, 1, 1, 800, 600], [2, 17, 33, 400, 300], [1, 100, 100, 100, 100]], np.float32)
    rois = np.concatenate([rois, good])
    out, arg = O.roi_pool(feat, rois, 7, 7, 1.0 / 16)
    assert np.array_equal(out, _roi_pool_numpy(feat, rois, 7, 7, 1.0 / 16))
    # argmax consistency: out == feat.flat[argmax] wherever the bin is non-empty
    for n in range(rois.shape[0]):
        b = int(rois[n, 0]) - 1
        for c in range(6):
            a = arg[n, c].ravel()
            v = out[n, c].ravel()
            ok = a >= 0
            assert np.array_equal(v[ok], feat[b, c].ravel()[a[ok]])
            assert np.all(v[~ok] == 0)


def test_foveal_context(O):
    rng = np.random.default_rng(6)
    rois = np.concatenate([np.ones((30, 1)), rng.uniform(1, 900, (30, 4))], 1).astype(np.float32)
    rois[:, 3:] = rois[:, 1:3] + rng.uniform(2, 300, (30, 2)).astype(np.float32)
    out = O.foveal(rois).reshape(30, 4, 5)
    r = rois.astype(np.float64)
    w, h = r[:, 3] - r[:, 1], r[:, 4] - r[:, 2]
    assert np.array_equal(out[:, 0], rois)
    for k, (off, mul) in enumerate([(0.25, 1.5), (0.5, 2.0), (1.5, 4.0)], start=1):  # Foveal.lua:37-39
        x, y = r[:, 1] - w * off, r[:, 2] - h * off
        exp = np.stack([r[:, 0], x, y, x + w * mul, y + h * mul], 1).astype(np.float32)
        assert np.array_equal(out[:, k], exp)
    # ContextRegion(1.5) is centre-preserving scale by 1.5 == Foveal row 2 up to fp32 rounding
    ctx = O.context_region(rois, 1.5)
    assert np.abs(ctx - out[:, 1]).max() < 1e-3
    assert np.array_equal(O.context_region(rois, 1.0), rois)


def test_transformer_and_projection(O):
    rng = np.random.default_rng(7)
    im = rng.random((3, 5, 9), dtype=np.float32)
    out = O.image_transform(im, **O.ROSS)
    exp = (im[[2, 1, 0]].astype(np.float64) * 255 - np.array(O.ROSS["mean"])[:, None, None]).astype(np.float32)
    assert np.array_equal(out, exp)
    assert O.pick_scale(600, 1000) == 1.0 and O.pick_scale(1000, 600) == 1.0
    assert O.pick_scale(480, 640) == 600 / 480 and O.pick_scale(300, 1000) == 1.0  # capped by max_size
    b = np.array([[1, 1, 100, 50], [3.5, 7.25, 1000, 600]], np.float32)
    assert np.array_equal(O.project_im_rois(b, 1.0)[:, 1:], b)
    assert np.array_equal(O.project_im_rois(b, 1.0)[:, 0], [1, 1])


def test_keep_top_k(O):
    rng = np.random.default_rng(8)
    per = [np.concatenate([rng.random((k, 4)), np.round(rng.random((k, 1)) * 50) / 50], 1).astype(np.float32) for k in (30, 0, 90, 5)]
    kept, t = O.keep_top_k(per, 100)
    alls = np.sort(np.concatenate([p[:, 4] for p in per]))[::-1]
    assert t == alls[99]
    assert sum(k.shape[0] for k in kept) == int((alls >= t).sum()) >= 100
    kept2, t2 = O.keep_top_k(per[:2], 100)  # fewer than k boxes: everything survives
    assert sum(k.shape[0] for k in kept2) == 30 and t2 == per[0][:, 4].min()
