"""ResNet Fast R-CNN (models/resnet.lua graph) through the C ABI vs the oracle: small networks with the
reference's topology (basic and bottleneck blocks, strided shortcuts, 7x7/2 stem, 3x3/2 max-pool, ROIPooling on the
stride-16 map, per-ROI layer4, average pool, cls/bbox heads).  Tolerance 1e-4 on scores (north_star), NMS results equal."""
import numpy as np
import pytest
import torch

from conftest import hooks

pytestmark = pytest.mark.gpu


def _inputs(H, W, N, seed):
    rng = np.random.default_rng(seed)
    im = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    c = rng.uniform([1, 1], [W, H], (N, 2))
    wh = np.exp(rng.uniform(np.log(8), np.log(min(H, W)), (N, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 1, W)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 1, H)
    return im, b.astype(np.float32)


@pytest.mark.parametrize("bt,blocks,width,pooled,N", [("bottleneck", [1, 1, 1, 1], 8, 14, 37), ("basic", [1, 2, 1, 2], 8, 14, 37), ("bottleneck", [2, 1, 2, 1], 16, 6, 37),
                                                   ("bottleneck", [1, 1, 1, 2], 16, 14, 150)])  # the last one is large enough for the 1x1 convolutions to take the GEMM path
@pytest.mark.parametrize("pf", [1, 0])  # 1 = LDS-DMA hand-pipelined convolution kernel for 32-channel-stage layers (default), 0 = the register-staged kernel
def test_resnet_frcnn_vs_oracle(O, dev, bt, blocks, width, pooled, N, pf):
    with hooks(fp32_pf=pf):
        _fp32_case(O, dev, bt, blocks, width, pooled, N)


def _fp32_case(O, dev, bt, blocks, width, pooled, N):
    from multipathnet_amd import models
    H, W, C = 97, 131, 6
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=width, blocks=blocks, block_type=bt, seed=21)
    Rn = models.resnet_params_numpy(R)
    im, boxes = _inputs(H, W, N, 4)
    net = models.ResNetFRCNN(R, pooled=pooled, max_h=H, max_w=W, max_rois=max(64, N), top_k=20)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    so, bo, logits, deltas = O.resnet_detect(im, boxes, Rn, target=min(H, W), max_size=max(H, W), pooled=pooled)  # s = 1, as the pipeline (no rescale configured)
    s, b = s.cpu().numpy(), b.cpu().numpy()
    assert np.abs(s - so).max() < 1e-4
    bo = O.clamp_boxes(bo.copy(), W, H)
    assert np.abs(b - bo).max() < 1e-2  # pixels; deltas are O(1e-3) * box size
    # cached features: the head alone on new boxes == a full run on them
    im2, boxes2 = _inputs(H, W, N, 5)
    s2, _ = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes2).to(dev), recompute_features=False)
    s2f, _ = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes2).to(dev))
    assert torch.equal(s2, s2f)


@pytest.mark.parametrize("bt,blocks,width,pooled,N", [("bottleneck", [1, 1, 1, 2], 16, 14, 150), ("basic", [1, 2, 1, 2], 8, 14, 37), ("bottleneck", [1, 1, 1, 3], 8, 6, 5)])
def test_resnet_head_3x3_on_the_winograd_mosaic(dev, bt, blocks, width, pooled, N):
    """graph_fuse bit 7: layer4's stride-1 3x3 convolutions run on the VGG pipeline's Winograd kernel over a MOSAIC image of the per-ROI maps
    (cells of (H + 1) x (W + 1) px, the spare row / column zero = the convolution's padding) instead of the generic per-pixel-tile
    kernel: another summation order, same result to rounding; also with fewer ROIs than the mosaic was laid out for, twice in a row"""
    from multipathnet_amd import models
    H, W, C = 97, 131, 6
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=width, blocks=blocks, block_type=bt, seed=23)
    im, boxes = _inputs(H, W, N, 5)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    out = []
    for fuse in (127, 255):
        with hooks(graph_fuse=fuse):
            net = models.ResNetFRCNN(R, pooled=pooled, max_h=H, max_w=W, max_rois=max(64, N), top_k=20)
            s, b = net.detect(imd, bd)
            s1 = s.cpu().numpy().copy()
            net.detect(imd, bd[: max(1, N // 3)])               # a smaller batch in between ...
            s2, _ = net.detect(imd, bd)                         # ... leaves nothing behind in the mosaic
            assert np.array_equal(s2.cpu().numpy(), s1)
            out.append((s1, b.cpu().numpy().copy()))
            del net
    assert not np.array_equal(out[0][0], out[1][0])             # really two code paths
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-5 and np.abs(out[0][1] - out[1][1]).max() < 1e-2


def test_resnet_test_one_pipeline(O, dev):
    """the whole Tester:testOne path (NMS, top-k, pipelined form) on the ResNet model: serial == pipelined exactly"""
    from multipathnet_amd import models
    H, W, N, C = 120, 160, 50, 5
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=8, blocks=[1, 1, 1, 1], block_type="bottleneck", seed=3)
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=10)
    im, boxes = _inputs(H, W, N, 9)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    n = int(net._n_dets.item())
    ref = net._dets[:n].clone()
    assert n > 0
    outs = [net.test_one_pipelined(imd, bd) for _ in range(3)]
    net.flush()
    torch.cuda.synchronize()
    for d, nd in outs:
        assert int(nd.item()) == n and torch.equal(d[:n], ref)


def test_resnet_multipathnet_extension_vs_oracle(O, dev):
    """BASELINE configs[3] shape (MultiPathNet on a ResNet backbone — defined by this library, the reference has no such model):
    Foveal towers over the stride-16 map, each with its own layer4 copy, K integral classifiers, box tower"""
    from multipathnet_amd import models
    H, W, N, C, K = 97, 131, 40, 5, 3
    R = models.synthetic_resnet_mpn_params(depth=0, n_classes=C, n_integral=K, base_width=8, blocks=[1, 1, 1, 1], block_type="bottleneck", seed=31)
    Rn = models.resnet_params_numpy(R)
    im, boxes = _inputs(H, W, N, 14)
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=20)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    so, bo = O.resnet_mpn_detect(im, boxes, Rn, target=min(H, W), max_size=max(H, W))
    assert np.abs(s.cpu().numpy() - so).max() < 1e-4
    assert np.abs(b.cpu().numpy() - O.clamp_boxes(bo, W, H)).max() < 1e-2
    net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    torch.cuda.synchronize()
    assert int(net._n_dets.item()) > 0


def test_resnet_iterative_localisation_and_voting_run(dev):
    """Tester_FRCNN.lua:82-99,118-124 on the ResNet model: the second localisation pass re-runs only the per-ROI head on the
    cached stride-16 map; deterministic run after run, and a later image of another size is handled"""
    from multipathnet_amd import models
    H, W, N, C = 120, 160, 48, 5
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=8, blocks=[1, 1, 1, 1], block_type="bottleneck", seed=5)
    im, boxes = _inputs(H, W, N, 2)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=10, num_iter=2, bbox_voting=True)
    net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    n = int(net._n_dets.item())
    assert n > 0
    d1 = net._dets[:n].clone()
    net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    assert int(net._n_dets.item()) == n and torch.equal(net._dets[:n], d1)
    # a different image size afterwards (buffers are sized for the maximum; the cached-feature check must see the new size)
    im2, boxes2 = _inputs(100, 140, N, 3)
    net.test_one_async(torch.from_numpy(im2).to(dev), torch.from_numpy(boxes2).to(dev))
    torch.cuda.synchronize()
    assert int(net._n_dets.item()) > 0


@pytest.mark.parametrize("bt,blocks,width,dma,tn", [("bottleneck", [1, 1, 1, 1], 16, 1, 0), ("basic", [1, 1, 1, 2], 16, 1, 0), ("bottleneck", [1, 1, 1, 1], 64, 1, 0),
                                                    ("bottleneck", [1, 1, 1, 1], 64, 2, 256), ("basic", [1, 1, 1, 2], 64, 2, 256),
                                                    ("bottleneck", [1, 1, 1, 1], 64, 2, 128), ("basic", [1, 1, 1, 2], 64, 2, 128),
                                                    ("bottleneck", [1, 1, 1, 1], 64, 2, 1256), ("basic", [1, 1, 1, 2], 64, 2, 1256),
                                                    ("bottleneck", [1, 1, 1, 1], 64, 2, 2256), ("basic", [1, 1, 1, 2], 64, 2, 2256)])
def test_resnet_bf16_vs_bf16_oracle_and_fp32(O, dev, bt, blocks, width, dma, tn):
    """bf16 graph (bf16 activations / weights, fp32 accumulate): against the oracle run with the same roundings (weights, the
    transformed image and every layer output rounded to bf16) the scores agree to bf16 accumulation-order noise; against
    the fp32 oracle they agree to bf16 precision.  width 64 exercises the 32-channel-per-stage kernel variant (widths 16: 16-channel);
    dma=2 forces the 256-cout LDS-DMA kernel onto every eligible layer (by default only layers with >= 32768 output pixels), tn its
    tile shape (0 = picked per layer, 128 / 256 = 256 couts x that many pixels, 1256 = 128 couts x 256 pixels, 2256 = the eight-wave 256 x 256
    shape of the debug flavour, where the cout count is a multiple of 256)."""
    with hooks(bf16_dma=dma, bf16_dma_tn=tn, bf16_bdir=0):   # (the B-direct kernel that large layers take by default: next test)
        _bf16_case(O, dev, bt, blocks, width)


@pytest.mark.parametrize("bt,blocks,width", [("bottleneck", [1, 1, 1, 1], 64), ("basic", [1, 1, 1, 2], 64)])
def test_resnet_bf16_b_direct_kernel_forced_and_bit_identical(O, dev, bt, blocks, width):
    """conv2d_c8i_bf16_bdir_kernel (round 4: weights through LDS, the pixel fragments straight into registers by buffer loads whose
    out-of-range offsets stand for the zero padding) forced onto every eligible layer — 1x1, 3x3 with padding, stride 2, residual
    epilogue, ragged pixel tiles: parity with the same-roundings oracle, and the SAME BITS as the LDS-DMA kernel (one accumulation
    chain per output in the same K order) — which is what lets either kernel serve a ROI batch of any size."""
    from multipathnet_amd import models
    outs = []
    for bdir, ver in ((2, 1), (0, 1), (2, 8)):   # ver 8: the 8-wave / 256-cout form (debug flavour only: measured slower, kept for the record)
        with hooks(bf16_dma=2, bf16_bdir=bdir, bf16_bdir_ver=ver):
            if bdir == 2 and ver == 1:
                _bf16_case(O, dev, bt, blocks, width)
            H, W, N, C = 97, 131, 37, 6
            R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=width, blocks=blocks, block_type=bt, seed=23)
            im, boxes = _inputs(H, W, N, 6)
            net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=20, bf16=True)
            s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
            outs.append((s.clone(), b.clone()))
            del net
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


def _bf16_case(O, dev, bt, blocks, width):
    from multipathnet_amd import models
    H, W, N, C = 97, 131, 37, 6
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=width, blocks=blocks, block_type=bt, seed=23)
    Rn = models.resnet_params_numpy(R)
    im, boxes = _inputs(H, W, N, 6)
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=20, bf16=True)
    s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    s = s.cpu().numpy()
    Rb = dict(Rn, bf16=True)
    sb, _, _, _ = O.resnet_detect(im, boxes, Rb, target=min(H, W), max_size=max(H, W))
    sf, _, _, _ = O.resnet_detect(im, boxes, Rn, target=min(H, W), max_size=max(H, W))
    assert np.abs(s - sb).max() < 3e-3          # same roundings: only fp32 summation order + rare 1-ulp bf16 flips differ
    assert np.abs(s - sf).max() < 3e-2          # bf16 vs fp32 arithmetic
    assert np.abs(s.sum(1) - 1).max() < 1e-5


@pytest.mark.parametrize("bf16", [True, False])
def test_resnet_fast_pooling_is_bit_identical(dev, bf16):
    """the row-per-thread ROI max-pooling (bf16: on the int16-sortable re-coding of the map) and the LDS average pooling of the
    bf16 graph reproduce the plain kernels' bits (max commutes with the monotone re-coding; the average sums in the same order)"""
    from multipathnet_amd import models
    H, W, N, C = 97, 131, 37, 6
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=16, blocks=[1, 1, 1, 1], block_type="bottleneck", seed=5)
    im, boxes = _inputs(H, W, N, 9)
    boxes[0, :] = [120.0, 90.0, 121.0, 91.0]      # a tiny ROI: bins narrower than a feature cell
    boxes[1, :] = [-40.0, -30.0, 10.0, 12.0]      # partly outside the image: empty bins -> 0
    out = []
    for fast in (0, 1, 2, 3):  # bit 0: ROI pooling, bit 1: average pooling; 3 = the product library
        with hooks(bf16_fast_pool=fast):
            net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=20, bf16=bf16)
            s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
            out.append((s.cpu().numpy().copy(), b.cpu().numpy().copy()))
            del net
    for k in (1, 2, 3):
        assert np.array_equal(out[0][0], out[k][0]) and np.array_equal(out[0][1], out[k][1]), k


@pytest.mark.parametrize("pooled", [16, 14])
def test_resnet_rows_do_not_depend_on_batch_for_even_head_maps(dev, pooled):
    """pooled 16 gives 8x8 head maps, a mosaic cell pitch of 9 — the Winograd tile phase of a ROI would then depend on its index in
    the batch.  Such sizes leave the mosaic route when rows must be batch-invariant: chunks, a ragged range and a permutation give bit-identical
    rows (memoryEfficientForward's property, ImageDetect.lua:126-133), as they do for the 7x7 maps (pitch 8) that stay on the mosaic."""
    from multipathnet_amd import models
    H, W, C, N = 97, 131, 6, 61
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=16, blocks=[1, 1, 1, 3], block_type="bottleneck", seed=29)
    im, boxes = _inputs(H, W, N, 7)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    net = models.ResNetFRCNN(R, pooled=pooled, max_h=H, max_w=W, max_rois=64, top_k=20)
    s, b = [t.clone() for t in net.detect(imd, bd)]
    for lo, hi in [(0, 7), (7, 30), (N - 1, N), (5, 6), (3, 3 + 40)]:
        s2, b2 = net.detect(imd, bd[lo:hi].contiguous(), recompute_features=False)
        assert torch.equal(s2, s[lo:hi]) and torch.equal(b2, b[lo:hi]), (pooled, lo, hi)
    perm = torch.from_numpy(np.random.default_rng(2).permutation(N)).to(dev)
    sp, bp = net.detect(imd, bd[perm].contiguous(), recompute_features=False)
    assert torch.equal(sp, s[perm]) and torch.equal(bp, b[perm])
