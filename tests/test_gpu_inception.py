"""Inception-v3 Fast R-CNN (models/inceptionv3.lua:27-43, BASELINE configs[4]) through mpn_graph_create vs the oracle's op-list
executor: a width-scaled network with the full module structure (InceptionA x3, B, C x4 in the trunk; D, E x2 per ROI; 1x7 / 7x1 /
1x3 / 3x1 kernels, average-pool branches, DepthConcat as side-by-side writes), fp32 and bf16."""
import numpy as np
import pytest
import torch

from conftest import hooks

pytestmark = pytest.mark.gpu


def _inputs(H, W, N, seed):
    rng = np.random.default_rng(seed)
    im = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(min(H, W)), (N, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 1, W)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 1, H)
    return im, b.astype(np.float32)


@pytest.mark.parametrize("bf16", [False, True])
def test_inception_frcnn_vs_oracle(O, dev, bf16):
    from multipathnet_amd import models
    H, W, N, C = 170, 215, 24, 5
    G = models.synthetic_inception_v3_params(n_classes=C, width=0.125, seed=13)
    Gn = models.graph_params_numpy(G)
    if bf16:
        Gn["bf16"] = True
    im, boxes = _inputs(H, W, N, 8)
    net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=32, top_k=10, bf16=bf16)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    so, bo, _, _ = O.graph_detect(im, boxes, Gn, O.INCEPTION, target=min(H, W), max_size=max(H, W))
    s = s.cpu().numpy()
    assert np.abs(s - so).max() < (3e-3 if bf16 else 1e-4)
    assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < (0.5 if bf16 else 1e-2)
    net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    torch.cuda.synchronize()
    assert int(net._n_dets.item()) > 0


def test_inception_multipathnet_extension_vs_oracle(O, dev):
    """BASELINE configs[4] shape: Foveal towers over the Inception trunk, each with its own Mixed_7a..7c copy, K integral
    classifiers, box tower (this library's extension), in bf16"""
    from multipathnet_amd import models
    H, W, N, C, K = 170, 215, 20, 4, 2
    G = models.synthetic_inception_mpn_params(n_classes=C, n_integral=K, width=0.125, seed=17, regions=[0, 2, 3, 1])
    Gn = dict(models.graph_params_numpy(G), bf16=True)
    im, boxes = _inputs(H, W, N, 12)
    net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=32, top_k=10, bf16=True)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    so, bo = O.graph_mpn_detect(im, boxes, Gn, O.INCEPTION, target=min(H, W), max_size=max(H, W))
    assert np.abs(s.cpu().numpy() - so).max() < 3e-3
    assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < 0.5


@pytest.mark.parametrize("tn", [0, 1256, 256, -1])
def test_inception_bf16_lds_dma_kernel_forced(O, dev, tn):
    """the LDS-DMA convolution kernel forced onto every eligible layer of a quarter-width Inception-v3 (by default only layers
    with >= 32768 output pixels use it): 1x7 / 7x1 / 1x3 / 3x1 taps, stride-2 reductions, DepthConcat slices as outputs, cout
    counts that are not multiples of the tile; tn = forced tile shape (0 = per layer, 1256 = 128 couts x 256 pixels, 256 = 256 x 256)"""
    from multipathnet_amd import models
    H, W, N, C = 170, 215, 24, 5
    G = models.synthetic_inception_v3_params(n_classes=C, width=0.25, seed=29)
    Gn = dict(models.graph_params_numpy(G), bf16=True)
    im, boxes = _inputs(H, W, N, 18)
    # tn = -1: the B-direct kernel (round 4) forced onto every eligible layer instead of the LDS-DMA one
    with hooks(bf16_dma=2, bf16_dma_tn=max(tn, 0), bf16_bdir=2 if tn < 0 else 0):
        net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=32, top_k=10, bf16=True)
        s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
        s = s.cpu().numpy()
        b = b.cpu()
    so, bo, _, _ = O.graph_detect(im, boxes, Gn, O.INCEPTION, target=min(H, W), max_size=max(H, W))
    assert np.abs(s - so).max() < 3e-3
    assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < 0.5


@pytest.mark.parametrize("bf16", [False, True])
def test_inception_sibling_fusion_equivalent(dev, bf16):
    """graph_parse fuses pointwise convolutions that read the same tensor into one convolution whose output the branches view
    by channel planes (bit 0), and commutes average-pool -> pointwise convolution (bit 1: the pool then runs on the convolution's
    output channels; exact in real arithmetic, another summation / rounding order in floating point); the same network built
    with the rewrites off (mpn_debug_set_graph_fuse(0)) gives the same scores to rounding"""
    from multipathnet_amd import models
    H, W, N, C = 170, 215, 24, 5
    G = models.synthetic_inception_v3_params(n_classes=C, width=0.25, seed=31)
    im, boxes = _inputs(H, W, N, 21)
    out = []
    for fuse in (3, 0, 1, 7, 71):
        with hooks(graph_fuse=fuse):  # 3 / 7 / 71 = subsets of the product library's rewrites (127): 3 + the max-pool of the ROI-pooled input computed from the feature map (+ 64: from range-max tables)
            net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=32, top_k=10, bf16=bf16)
            s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
            out.append((s.cpu().numpy().copy(), b.cpu().numpy().copy()))
            del net
    assert np.abs(out[2][0] - out[1][0]).max() < (1e-3 if bf16 else 1e-6)   # sibling fusion alone: the same dot products
    assert np.abs(out[2][1] - out[1][1]).max() < (0.25 if bf16 else 1e-3)
    assert np.abs(out[0][0] - out[1][0]).max() < (2e-3 if bf16 else 2e-5)   # + the commuted pools
    assert np.abs(out[0][1] - out[1][1]).max() < (0.5 if bf16 else 5e-3)
    # bit 2 (bf16 only; fp32 graphs ignore it): Mixed_7a's max-pool of the ROI-pooled input taken straight from the feature map — max of
    # maxes over the union of the bins' windows, exact: identical scores and boxes
    assert np.array_equal(out[3][0], out[0][0]) and np.array_equal(out[3][1], out[0][1])
    # bit 6: the same fused max-pool reading two rows of a vertical range-max table per window column instead of every row: still a max
    assert np.array_equal(out[4][0], out[0][0]) and np.array_equal(out[4][1], out[0][1])
