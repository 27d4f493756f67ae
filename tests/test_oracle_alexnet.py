"""BASELINE configs[0] — the oracle's restatement of models/alexnet.lua's graph (grouped convolutions, cross-channel LRN, ceil-mode
3x3/2 max-pooling, fc6 / fc7 on the flattened 6x6 ROI-pooled map) cross-checked against PyTorch-CPU.  The arithmetic lives in the
external nn / cudnn rocks and the layer list in an absent `.t7` (PARITY UNPINNED): PyTorch's conv2d(groups) / max_pool2d(ceil_mode) /
local_response_norm / linear are the independent statement of the same published semantics."""
import numpy as np
import torch
import torch.nn.functional as F


def _torch_trunk(G, x):
    ops = G["trunk_ops"]
    t = torch.from_numpy(x)
    i = 0
    while i < len(ops):
        o = ops[i]
        if o["kind"] == 0 and o.get("src_off") is not None:   # a grouped convolution: its ops are consecutive, one per group
            grp = [o]
            while i + 1 < len(ops) and ops[i + 1]["kind"] == 0 and ops[i + 1]["dst"] == o["dst"]:
                i += 1
                grp.append(ops[i])
            w, b = torch.cat([g["w"] for g in grp]), torch.cat([g["b"] for g in grp])
            t = F.relu(F.conv2d(t, w, b, stride=o["sh"], padding=o["ph"], groups=len(grp)))
        elif o["kind"] == 0:
            t = F.relu(F.conv2d(t, o["w"], o["b"], stride=o["sh"], padding=o["ph"]))
        elif o["kind"] == 1:
            t = F.max_pool2d(t, o["kh"], o["sh"], o["ph"], ceil_mode=bool(o.get("ceil")))
        elif o["kind"] == 3:
            t = F.local_response_norm(t, o["kh"], o["lrn"][0], o["lrn"][1], o["lrn"][2])
        i += 1
    return t


def test_alexnet_trunk_vs_pytorch(O):
    from multipathnet_amd import models
    for (h, w, width) in [(75, 131, 0.25), (97, 64, 0.5), (227, 227, 0.25)]:
        G = models.synthetic_alexnet_params(n_classes=5, width=width, fc_dim=64, seed=h)
        Gn = models.graph_params_numpy(G)
        x = (np.random.default_rng(h).random((1, 3, h, w), dtype=np.float32) * 255 - 110)
        feat = O.graph_run(x, Gn["trunk_ops"], Gn["trunk_tensor_c"])[Gn["feat_tensor"]]
        ref = _torch_trunk(G, x).numpy()
        assert feat.shape == ref.shape
        assert np.abs(feat - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_pool_ceil_sizes_and_lrn_vs_pytorch(O):
    rng = np.random.default_rng(1)
    for h in (6, 7, 13, 27, 55, 150, 151):
        x = rng.standard_normal((2, 8, h, h + 3)).astype(np.float32)
        for (k, s, p) in [(3, 2, 1), (3, 2, 0), (2, 2, 0)]:
            a = O.maxpool2d_mode(x, k, s, p, 1)
            b = F.max_pool2d(torch.from_numpy(x), k, s, p, ceil_mode=True).numpy()
            assert a.shape == b.shape and np.array_equal(a, b), (h, k, s, p)
    x = rng.standard_normal((2, 40, 9, 11)).astype(np.float32) * 30
    for size in (3, 5, 9):
        a = O.lrn(x, size, 1e-4, 0.75, 1.0)
        b = F.local_response_norm(torch.from_numpy(x), size, 1e-4, 0.75, 1.0).numpy()
        assert np.abs(a - b).max() < 1e-5 * np.abs(b).max()


def test_alexnet_head_is_linear_on_the_flattened_pool(O):
    """`top` (alexnet.lua:24-25): nn.View(-1) + Linear(256*6*6, 4096) == the 6x6 convolution of the op list on the [256,6,6] map"""
    from multipathnet_amd import models
    G = models.synthetic_alexnet_params(n_classes=4, width=0.25, fc_dim=32, seed=2)
    Gn = models.graph_params_numpy(G)
    c = Gn["head_tensor_c"][0]
    x = np.random.default_rng(3).standard_normal((7, c, 6, 6)).astype(np.float32)
    y = O.graph_run(x, Gn["head_ops"], Gn["head_tensor_c"])[Gn["out_tensor"]]
    h = torch.from_numpy(x.reshape(7, -1))
    h = F.relu(F.linear(h, G["head_ops"][0]["w"].reshape(32, -1), G["head_ops"][0]["b"]))
    h = F.relu(F.linear(h, G["head_ops"][1]["w"].reshape(32, -1), G["head_ops"][1]["b"]))
    assert y.shape == (7, 32, 1, 1) and np.abs(y[:, :, 0, 0] - h.numpy()).max() < 1e-5
