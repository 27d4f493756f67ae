"""fp32-MFMA conv / linear / pool through the module-level C ABI vs the oracle (tolerance 1e-4 relative
to the output scale, north_star) and the reference's exactness properties (chunked == un-chunked)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import hooks

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _conv(dev, x, w, b, relu=True):
    from multipathnet_amd import nn
    m = nn.SpatialConvolution(w.shape[1], w.shape[0], relu=relu)
    m.weight, m.bias = _t(w, dev), (_t(b, dev) if b is not None else None)
    return m.forward(_t(x, dev)).cpu().numpy()


@pytest.mark.parametrize("ci,co,h,w", [(3, 64, 40, 70), (8, 64, 33, 31), (64, 64, 19, 45), (64, 128, 16, 96), (128, 256, 9, 33),
                                       (24, 40, 8, 8), (16, 200, 5, 37), (256, 512, 12, 20)])
@pytest.mark.parametrize("variant,split", [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (5, 2), (6, 3), (1, 2), (2, 3), (3, 2), (4, 3), (0, 4),
                                           (7, 0), (7, 2), (7, 3), (7, -2), (7, -3)])
def test_conv3x3_vs_oracle(O, dev, ci, co, h, w, variant, split):
    """variant 0 = the default choice per layer, 1-6 = direct-convolution tilings, 7 = Winograd F(2x2,3x3); split = forced split-K factor
    (0 = cost model; negative = a TAIL split of that many K ranges over the second half of the tiles, the first half un-split)"""
    rng = np.random.default_rng(ci * 1000 + co)
    x = rng.standard_normal((ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * (2.0 / (ci * 9)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    with hooks(conv_variant=variant, conv_split=split):  # (0, 0) = the product library's own dispatch
        y = _conv(dev, x, wt, b, relu=True)
        y2 = _conv(dev, x, wt, None, relu=False)
    ref = O.conv3x3(x, wt, b, relu=True)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(y2 - O.conv3x3(x, wt, None, relu=False)).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ci,co,h,w", [(64, 64, 19, 45), (128, 256, 9, 33), (16, 200, 5, 37), (256, 512, 12, 20), (24, 40, 38, 63), (32, 64, 75, 125)])
@pytest.mark.parametrize("tc,split", [(8, 0), (16, 0), (16, 2), (16, -2), (8, -3)])
def test_conv3x3_winograd_block_geometries(O, dev, ci, co, h, w, tc, split):
    """the Winograd kernel's two block geometries (tc = 8: 16 x 16 output px, tc = 16: 8 rows x 32 columns; the host picks per layer
    whichever pads the map less) with uniform / tail split-K, bias + ReLU and the module-level path's layouts"""
    rng = np.random.default_rng(ci * 31 + co + tc)
    x = rng.standard_normal((ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * (2.0 / (ci * 9)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    with hooks(conv_variant=7, conv_split=split, wino_tc=tc):
        y = _conv(dev, x, wt, b, relu=True)
    ref = O.conv3x3(x, wt, b, relu=True)
    assert np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_conv_transpose_detecting(O, dev):
    """asymmetric weights + a single hot pixel: catches row/col or tap transposes (guide §5.4 rule 16)"""
    x = np.zeros((8, 12, 40), np.float32)
    x[3, 5, 17] = 1.0
    wt = np.arange(16 * 8 * 9, dtype=np.float32).reshape(16, 8, 3, 3) / 100.0
    ref = O.conv3x3(x, wt, None, relu=False)
    y = _conv(dev, x, wt, None, relu=False)  # default: Winograd (fp32 transforms -> not bit-exact; adjacent weights differ by 1e-2)
    assert np.abs(y - ref).max() < 1e-4
    with hooks(conv_variant=1):  # direct kernel: one product per output -> exact
        assert np.array_equal(_conv(dev, x, wt, None, relu=False), ref)


def test_maxpool_ceil_exact(O, dev):
    from multipathnet_amd import nn
    rng = np.random.default_rng(1)
    for (c, h, w) in [(64, 75, 125), (8, 38, 63), (3, 1, 1), (5, 2, 3), (2, 600, 11)]:
        x = rng.standard_normal((c, h, w)).astype(np.float32)
        assert np.array_equal(nn.SpatialMaxPooling().forward(_t(x, dev)).cpu().numpy(), O.maxpool2x2_ceil(x))


def _linear(dev, x, w, b, relu=False):
    from multipathnet_amd import nn
    m = nn.Linear(w.shape[1], w.shape[0], relu=relu)
    m.weight, m.bias = _t(w, dev), (_t(b, dev) if b is not None else None)
    return m.forward(_t(x, dev))


@pytest.mark.parametrize("M,K,N", [(1, 8, 1), (40, 512, 9), (130, 300, 70), (257, 4096, 105), (1000, 1024, 512), (64, 25088, 128)])
def test_linear_vs_oracle(O, dev, M, K, N):
    import multipathnet_amd
    lib = multipathnet_amd.load()
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * (1.0 / K) ** 0.5).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    y = _linear(dev, x, w, b, relu=True).cpu().numpy()
    ref = O.linear(x, w, b, relu=True)
    assert np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_linear_asymmetric_identity(O, dev):
    K = N = 64
    x = np.arange(5 * K, dtype=np.float32).reshape(5, K)
    w = np.eye(N, K, dtype=np.float32)
    w[3, 7] = 2.0  # asymmetric
    y = _linear(dev, x, w, None).cpu().numpy()
    assert np.array_equal(y, O.linear(x, w, None))


def test_split_batch_invariance_exact(O, dev):
    """test.lua:140-179 (SequentialSplitBatch): ROIPooling+Linear and plain Linear, chunked (25) output ==
    un-chunked output EXACTLY (asserteq ... 0)."""
    from multipathnet_amd import nn
    rng = np.random.default_rng(3)
    feat = _t(rng.standard_normal((1, 512, 38, 50)).astype(np.float32), dev)
    rois = (rng.standard_normal((40, 5)) * 50).astype(np.float32)
    rois[:, 0] = 1
    rois = _t(rois, dev)
    w = (rng.standard_normal((9, 7 * 7 * 512)) * 0.01).astype(np.float32)
    b = rng.standard_normal(9).astype(np.float32)
    pool = nn.ROIPooling(7, 7, 1 / 16)
    run = lambda r: _linear(dev, pool.forward([feat, r]).reshape(r.size(0), -1).cpu().numpy(), w, b).clone()
    full = run(rois)
    parts = torch.cat([run(rois[:25].contiguous()), run(rois[25:].contiguous())])
    assert (full - parts).abs().max().item() == 0
    x = rng.standard_normal((40, 512)).astype(np.float32)
    w2 = rng.standard_normal((9, 512)).astype(np.float32)
    full = _linear(dev, x, w2, b)
    parts = torch.cat([_linear(dev, x[:25], w2, b).clone(), _linear(dev, x[25:], w2, b).clone()])
    assert (full - parts).abs().max().item() == 0


@pytest.mark.parametrize("M,K,N", [(130, 300, 70), (1000, 1024, 512), (64, 25088, 128), (257, 4096, 105), (5, 8, 3)])
def test_linear_repeatable(O, dev, M, K, N):
    """the hand-pipelined LDS-DMA GEMM gives bit-identical results run after run (screens for DMA / read races) and matches the oracle"""
    rng = np.random.default_rng(M * 7 + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * (1.0 / K) ** 0.5).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    base = _linear(dev, x, w, b, relu=True).clone()
    for _ in range(5):
        assert torch.equal(_linear(dev, x, w, b, relu=True), base)
    assert np.abs(base.cpu().numpy() - O.linear(x, w, b, relu=True)).max() < 1e-4 * max(1.0, float(base.abs().max()))


@pytest.mark.parametrize("variant", [1, 7])
@pytest.mark.parametrize("ci,co,h,w", [(72, 200, 75, 125), (64, 64, 150, 200), (256, 96, 40, 333)])
def test_conv_splitk_deterministic(O, dev, ci, co, h, w, variant):
    """split-K over the input-channel chunks (fp32 slabs summed in a fixed order): bit-identical run after run, equal to the unsplit
    kernel within fp32 reassociation, and within the north_star tolerance of the oracle; direct (1) and Winograd (7) kernels"""
    rng = np.random.default_rng(ci + co)
    x = rng.standard_normal((ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * (2.0 / (ci * 9)) ** 0.5).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    with hooks(conv_variant=variant, conv_split=3):
        a = _conv(dev, x, wt, b)
        for _ in range(3):
            assert np.array_equal(_conv(dev, x, wt, b), a)
    with hooks(conv_variant=variant, conv_split=1):
        c = _conv(dev, x, wt, b)
    ref = O.conv3x3(x, wt, b, relu=True)
    assert np.abs(a - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(a - c).max() < 1e-4 * max(1.0, np.abs(ref).max())


