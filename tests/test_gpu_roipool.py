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
