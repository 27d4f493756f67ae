"""ResNet-50 Fast R-CNN at FULL size (600x1000 image, 1000 ROIs — the shapes tools/bench_resnet.py times, where the large-layer
kernels are the ones that run: the LDS-DMA bf16 convolution in all its tile shapes, the LDS-DMA fp32 convolution, the n-fast GEMM
tile order, the row-per-thread ROI pooling): parity against the oracle on a ROI sample.  The oracle runs the whole trunk on the
host cores and layer4 for 12 of the 1000 ROIs; ROIs are independent, so the device's rows for those ROIs must match."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16,N", [(False, 1000), (True, 1000), (True, 173), (False, 173)])  # 173 ROIs: 33 908 pooled pixels, just past the large-layer threshold, ragged tiles
def test_resnet50_fullsize_scores_vs_oracle_on_roi_sample(O, dev, bf16, N):
    import bench
    from multipathnet_amd import models
    H, W = bench.H, bench.W
    R = models.synthetic_resnet_params(depth=50, n_classes=21, seed=91)
    Rn = models.resnet_params_numpy(R)
    if bf16:
        Rn = dict(Rn, bf16=True)
    im, boxes = bench.synthetic_inputs()
    boxes = np.ascontiguousarray(boxes[:N])
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=1000, bf16=bf16)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s, b = s.cpu().numpy(), b.cpu().numpy()
    assert np.isfinite(s).all() and np.abs(s.sum(1) - 1).max() < 1e-5
    idx = np.random.default_rng(11).choice(N, 12, replace=False)
    idx[0] = N - 1  # the last ROI: the ragged end of the last pixel tile
    so, bo, _, _ = O.resnet_detect(im, boxes[idx], Rn, target=min(H, W), max_size=max(H, W))  # scale 1, as the pipeline
    assert np.abs(s[idx] - so).max() < (3e-3 if bf16 else 1e-4)
    assert np.abs(b[idx] - O.clamp_boxes(bo.copy(), W, H)).max() < (0.5 if bf16 else 1e-2)
    # determinism of the full-size path
    s2, b2 = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    assert np.array_equal(s2.cpu().numpy(), s) and np.array_equal(b2.cpu().numpy(), b)


@pytest.mark.parametrize("bf16", [False, True])
def test_inception_v3_fullsize_scores_vs_oracle_on_roi_sample(O, dev, bf16):
    """BASELINE configs[4]'s backbone at full size (600x1000, 2000 ROIs, full width): the fused sibling convolutions, every LDS-DMA
    tile shape the host picks, the LDS pooling kernels — against the oracle's op-list executor on 8 of the 2000 ROIs"""
    import bench
    from multipathnet_amd import models
    H, W, N, C = bench.H, bench.W, 2000, 21
    G = models.synthetic_inception_v3_params(n_classes=C, width=1.0, seed=77)
    Gn = models.graph_params_numpy(G)
    if bf16:
        Gn = dict(Gn, bf16=True)
    rng = np.random.default_rng(5)
    im = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(min(H, W)), (N, 2)))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    boxes[:, [0, 2]] = np.clip(boxes[:, [0, 2]], 1, W)
    boxes[:, [1, 3]] = np.clip(boxes[:, [1, 3]], 1, H)
    boxes = boxes.astype(np.float32)
    net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=N, bf16=bf16)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s, b = s.cpu().numpy(), b.cpu().numpy()
    assert np.isfinite(s).all() and np.abs(s.sum(1) - 1).max() < 1e-5
    idx = rng.choice(N, 8, replace=False)
    so, bo, _, _ = O.graph_detect(im, boxes[idx], Gn, O.INCEPTION, target=min(H, W), max_size=max(H, W))
    assert np.abs(s[idx] - so).max() < (3e-3 if bf16 else 1e-4)
    assert np.abs(b[idx] - O.clamp_boxes(bo.copy(), W, H)).max() < (0.5 if bf16 else 1e-2)


def test_resnet50_multipathnet_fullsize_vs_oracle_on_roi_sample(O, dev):
    """BASELINE configs[3] at full size (ResNet-50, 5 Foveal towers, K = 6, 81 classes, 1000 ROIs) in fp32: 6 sampled ROIs"""
    import bench
    from multipathnet_amd import models
    H, W, N = bench.H, bench.W, bench.N_ROIS
    R = models.synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=93)
    Rn = models.resnet_params_numpy(R)
    im, boxes = bench.synthetic_inputs()
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=N)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s, b = s.cpu().numpy(), b.cpu().numpy()
    idx = np.random.default_rng(13).choice(N, 6, replace=False)
    so, bo = O.resnet_mpn_detect(im, boxes[idx], Rn, target=min(H, W), max_size=max(H, W))
    assert np.abs(s[idx] - so).max() < 1e-4
    assert np.abs(b[idx] - O.clamp_boxes(bo.copy(), W, H)).max() < 1e-2


def test_inception_multipathnet_fullsize_vs_oracle_on_roi_sample(O, dev):
    """BASELINE configs[4] at full size (Inception-v3, 5 Foveal towers, K = 6, 81 classes, 2000 ROIs, bf16): 4 sampled ROIs"""
    import bench
    from multipathnet_amd import models
    H, W, N = bench.H, bench.W, 2000
    G = models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=95)
    Gn = dict(models.graph_params_numpy(G), bf16=True)
    rng = np.random.default_rng(6)
    im = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(min(H, W)), (N, 2)))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    boxes[:, [0, 2]] = np.clip(boxes[:, [0, 2]], 1, W)
    boxes[:, [1, 3]] = np.clip(boxes[:, [1, 3]], 1, H)
    boxes = boxes.astype(np.float32)
    net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=N, bf16=True)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s, b = s.cpu().numpy(), b.cpu().numpy()
    idx = rng.choice(N, 4, replace=False)
    so, bo = O.graph_mpn_detect(im, boxes[idx], Gn, O.INCEPTION, target=min(H, W), max_size=max(H, W))
    assert np.abs(s[idx] - so).max() < 3e-3
    assert np.abs(b[idx] - O.clamp_boxes(bo.copy(), W, H)).max() < 0.5
