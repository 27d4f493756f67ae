"""oracle/torch_ref.py (the PyTorch-CPU leg of the full-size graph-model parity tests) against the oracle on small networks, fp32 and
with the bf16 storage emulation: two independent CPU implementations of the same public definitions (models/resnet.lua:28-50,
models/inceptionv3.lua:27-43)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T


def _boxes(rng, n, W, H):
    c = rng.uniform([1, 1], [W, H], (n, 2))
    wh = np.exp(rng.uniform(np.log(12), np.log(min(H, W)), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 1, W)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 1, H)
    return b.astype(np.float32)


@pytest.mark.parametrize("bf16", [False, True])
def test_resnet_torch_ref_vs_oracle(O, bf16):
    from multipathnet_amd import models
    R = models.rescale_heads(models.synthetic_resnet_params(depth=0, n_classes=5, base_width=8, blocks=[1, 1, 2, 2], block_type="bottleneck", seed=3))
    Rn = dict(models.resnet_params_numpy(R), bf16=bf16)
    rng = np.random.default_rng(5)
    H, W, N = 96, 128, 9
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, N, W, H)
    _, _, lo, do = O.resnet_detect(im, boxes, Rn, target=H, max_size=W, pooled=6)
    feat = T.resnet_trunk(O.image_transform(im, **O.IMAGENET), R, bf16)
    pooled, _ = O.roi_pool(feat, O.project_im_rois(boxes, 1.0), 6, 6, 1.0 / 16)
    lt, dt = T.heads(T.resnet_tower(pooled, R["head_blocks"], bf16), R, 5)
    tol = 5e-2 if bf16 else 1e-4
    assert np.abs(lt - lo).max() < tol * max(1.0, np.abs(lo).max()) and np.abs(dt - do).max() < tol


@pytest.mark.parametrize("bf16", [False, True])
def test_inception_torch_ref_vs_oracle(O, bf16):
    from multipathnet_amd import models
    G = models.rescale_heads(models.synthetic_inception_v3_params(n_classes=4, width=0.125, seed=9))
    Gn = dict(models.graph_params_numpy(G), bf16=bf16)
    rng = np.random.default_rng(6)
    H, W, N = 150, 200, 5
    im = rng.random((3, H, W), dtype=np.float32)
    boxes = _boxes(rng, N, W, H)
    _, _, lo, do = O.graph_detect(im, boxes, Gn, O.INCEPTION, target=H, max_size=W)
    feat = T.graph_trunk(O.image_transform(im, **O.INCEPTION), G, bf16)
    pooled, _ = O.roi_pool(feat, O.project_im_rois(boxes, 1.0), 17, 17, 17.0 / 299.0)
    lt, dt = T.heads(T.graph_tower(pooled, G["head_ops"], G, bf16), G, 4)
    tol = 5e-2 if bf16 else 1e-4
    assert np.abs(lt - lo).max() < tol * max(1.0, np.abs(lo).max()) and np.abs(dt - do).max() < tol
