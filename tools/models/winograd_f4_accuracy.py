"""CPU study for DESIGN.md §9: what would Winograd F(4x4,3x3) cost the VGG-16 trunk in accuracy?

The device trunk runs F(2x2,3x3) in fp32 on every 3x3 layer but the first (dense.hip conv3x3_wino_kernel).  F(4x4,3x3) needs
36 multiplies per 16 outputs instead of 16 per 4 (1.78x fewer again), but its transform matrices hold 1/24 ... 8, so fp32
cancellation error grows.  This script runs the 13-layer trunk (synthetic_params weights, the bench's value ranges) on one image four
ways and reports the conv5_3 feature error against float64:
    direct fp32 (torch conv2d)  |  F(2x2) emulated in fp32  |  F(4x4) emulated in fp32  |  F(4x4) on conv1_2..conv3_3 only
The emulation follows the kernel's arithmetic: U = G g G^T in float64 rounded once to fp32 (pack_conv_w_wino_kernel), V = B^T d B and
the channel sum in fp32, Y = A^T M A in fp32.  No GPU, no oracle: numpy / torch-CPU only.   usage: python tools/models/winograd_f4_accuracy.py [H W]
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from multipathnet_amd import models  # noqa: E402

MATS = {
    2: dict(
        BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
        G=[[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]],
        AT=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: dict(
        BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
        G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
        AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def wino_conv(x, w, b, m):
    """x [C,H,W] fp32, w [K,C,3,3], pad 1 -> [K,H,W]; F(m x m, 3x3) with every stage in fp32"""
    mats = MATS[m]
    BT = torch.tensor(mats["BT"], dtype=torch.float32)
    AT = torch.tensor(mats["AT"], dtype=torch.float32)
    G64 = torch.tensor(mats["G"], dtype=torch.float64)
    U = (G64 @ w.double() @ G64.T).float()                      # [K,C,a,a], rounded once
    a = m + 2
    C, H, W = x.shape
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, 1 + tw * m - W, 1, 1 + th * m - H))
    tiles = xp.unfold(1, a, m).unfold(2, a, m)                  # [C,th,tw,a,a]
    V = BT @ tiles @ BT.T                                       # fp32
    M = torch.einsum("kcij,cyxij->kyxij", U, V)                 # fp32 channel sum
    Y = AT @ M @ AT.T                                           # [K,th,tw,m,m]
    y = Y.permute(0, 1, 3, 2, 4).reshape(w.shape[0], th * m, tw * m)[:, :H, :W]
    return y + b[:, None, None]


def trunk(x, P, mode, dtype=torch.float32):
    """mode: per-3x3-layer tile size (0 = direct conv2d); the first layer is always direct, as on the device"""
    li = 0
    for item in models.VGG16_CFG:
        if item == "P":
            x = F.max_pool2d(x[None], 2, 2, ceil_mode=True)[0]
            continue
        w, b = P["conv_w"][li].to(dtype), P["conv_b"][li].to(dtype)
        m = 0 if li == 0 else mode[li]
        x = F.conv2d(x[None], w, b, padding=1)[0] if m == 0 else wino_conv(x, w, b, m)
        x = torch.relu(x)
        li += 1
    return x


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 224)
    torch.manual_seed(0)
    torch.set_num_threads(32)
    P = models.synthetic_params(models.VGG16_CFG, seed=557)
    rng = np.random.default_rng(3)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32))
    x = im[[2, 1, 0]] * 255.0 - torch.tensor([102.9801, 115.9465, 122.7717])[:, None, None]   # the Ross transformer's ranges
    ref = trunk(x.double(), P, [0] * 13, torch.float64)
    scale = float(ref.abs().max())
    rms = float(ref.pow(2).mean().sqrt())
    print("image %dx%d; conv5_3 features: max %.3f, rms %.3f" % (H, W, scale, rms))
    cases = {
        "direct fp32": [0] * 13,
        "F(2x2) all 12 layers (the device trunk)": [2] * 13,
        "F(4x4) all 12 layers": [4] * 13,
        "F(4x4) conv1_2..conv3_3, F(2x2) conv4/5": [4] * 7 + [2] * 6,
        "F(4x4) conv4/5 only, F(2x2) before": [2] * 7 + [4] * 6,
    }
    for name, mode in cases.items():
        y = trunk(x, P, mode)
        e = (y.double() - ref).abs()
        print("  %-45s max abs err %.3e (%.2e of max)   rms err %.3e" % (name, float(e.max()), float(e.max()) / scale, float(e.pow(2).mean().sqrt())))


if __name__ == "__main__":
    main()
