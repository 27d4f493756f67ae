"""Model objects for the hot path (models/vgg.lua, models/alexnet.lua shape of graph).

`FastRCNN` owns one fused device pipeline (mpn_frcnn_* in include/mpn.h): trunk -> ROIPooling ->
fc6/fc7 -> {cls, bbox} (+BBoxNorm) -> softmax/decode/clamp -> per-class NMS -> top-k.  Weights are
given in Torch layout (conv [Cout,Cin,3,3], linear [out,in]) and re-packed once into HBM-resident
MFMA-fragment order.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import FrcnnConfig, GraphOp, GraphWeights, MpnetWeights, ResnetWeights, check, f32p
from .nn import _f, _i, _stream

VGG16_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512]  # vgg.lua:14-27, no pool5


def cfg_layers(cfg):
    cout, pool = [], []
    for item in cfg:
        if item == "P":
            pool[-1] = 1
        else:
            cout.append(int(item))
            pool.append(0)
    return cout, pool


HEAD_SCALES = {  # (cls_w std, cls_b std, bbox_w std, bbox_b std)
    "init": (0.01, 0.0, 0.001, 0.0),       # model_utils.lua:106-112: the INITIALISATION of classAndBBoxLinear
    "trained": (0.03, 1.0, 0.005, 0.1),    # magnitudes of a trained detector: class logits reach +-15, box deltas O(0.3)
}


def synthetic_params(cfg=VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557, device="cpu", bbox_norm=True, head_scale="init"):
    """Seeded random weights of the reference architecture (no pretrained .t7 exists offline):
    He-scaled trunk / fc so activations stay O(1) — the first conv is additionally divided by 70, the
    spread of the mean-subtracted 0..255 pixels it sees, like a trained Caffe VGG whose conv1_1 filters
    are O(1e-2); heads as model_utils.lua:106-112 (cls N(0,0.01), bbox N(0,0.001), zero bias).
    head_scale="trained": head weights / biases at the magnitude of a trained detector instead of the initialisation
    (the trunk and fc draws are the same for both, so the two parameter sets share every other tensor)."""
    cls_std, cls_bstd, bbox_std, bbox_bstd = HEAD_SCALES[head_scale]
    g = torch.Generator().manual_seed(seed)
    P = {"conv_w": [], "conv_b": []}
    cin = 3
    for item in cfg:
        if item == "P":
            continue
        std = (2.0 / (cin * 9)) ** 0.5
        if cin == 3:
            std /= 70.0
        P["conv_w"].append((torch.randn(item, cin, 3, 3, generator=g) * std).to(device))
        P["conv_b"].append((torch.randn(item, generator=g) * 0.01).to(device))
        cin = item
    k6 = cin * pooled * pooled
    P["fc6_w"] = (torch.randn(fc_dim, k6, generator=g) * (2.0 / k6) ** 0.5).to(device)
    P["fc6_b"] = (torch.randn(fc_dim, generator=g) * 0.01).to(device)
    P["fc7_w"] = (torch.randn(fc_dim, fc_dim, generator=g) * (2.0 / fc_dim) ** 0.5).to(device)
    P["fc7_b"] = (torch.randn(fc_dim, generator=g) * 0.01).to(device)
    P["cls_w"] = (torch.randn(n_classes, fc_dim, generator=g) * cls_std).to(device)
    P["bbox_w"] = (torch.randn(4 * n_classes, fc_dim, generator=g) * bbox_std).to(device)
    # biases are drawn AFTER the weights so that "init" consumes the generator exactly as before
    P["cls_b"] = (torch.randn(n_classes, generator=g) * cls_bstd).to(device)
    P["bbox_b"] = (torch.randn(4 * n_classes, generator=g) * bbox_bstd).to(device)
    if bbox_norm:  # typical Fast R-CNN target statistics (train.lua:136-138 adds the module)
        P["bbox_mean"] = [0.0, 0.0, 0.0, 0.0]
        P["bbox_std"] = [0.1, 0.1, 0.2, 0.2]
    else:
        P["bbox_mean"] = P["bbox_std"] = None
    return P


def rescale_heads(P, head_scale="trained", seed=7001, cls_gain=1.0):
    """A copy of ANY parameter dict of this module (VGG / MultiPathNet / ResNet / op-list, plain or tower form) whose cls / bbox head tensors
    are redrawn, same shapes, at HEAD_SCALES[head_scale] — e.g. "trained": the score scale of a trained detector (class logits of +-10 and
    more, non-zero biases, saturating softmax rows) instead of model_utils.lua:106-112's initialisation.  Every other tensor is shared.
    cls_gain multiplies the class weights' std (backbones whose pooled features are small need it to reach the same logit range)."""
    cls_std, cls_bstd, bbox_std, bbox_bstd = HEAD_SCALES[head_scale]
    g = torch.Generator().manual_seed(seed)
    Q = dict(P)
    Q["cls_w"] = torch.randn(P["cls_w"].shape, generator=g) * (cls_std * cls_gain)
    Q["bbox_w"] = torch.randn(P["bbox_w"].shape, generator=g) * bbox_std
    Q["cls_b"] = torch.randn(P["cls_b"].shape, generator=g) * cls_bstd
    Q["bbox_b"] = torch.randn(P["bbox_b"].shape, generator=g) * bbox_bstd
    return Q


# fb.resnet.torch topologies (the `.t7` models/resnet.lua:17,25 loads): blocks per layer and block type
RESNET_DEFS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3])}
IMAGENET_TRANSFORMER = dict(mean=(0.48462227599918, 0.45624044862054, 0.40588363755159),
                            std=(0.22889466674951, 0.22446679341259, 0.22495548344775), scale=1.0, swap=(0, 1, 2))  # model_utils.lua:143-155


def synthetic_resnet_params(depth=50, n_classes=21, seed=557, base_width=64, blocks=None, block_type=None, bbox_norm=True):
    """Seeded random ResNet Fast R-CNN weights with BatchNorm already folded into (w, b) per convolution
    (inn.utils.BNtoFixed, resnet.lua:34-36): conv1 7x7/2 -> [max-pool] -> layer1..3 (trunk) | layer4 (per-ROI head).
    basic block: 3x3(stride) , 3x3; bottleneck: 1x1, 3x3(stride), 1x1 x4; shortcut type B (strided 1x1 convolution where
    the shape changes).  He-scaled; the last convolution of every block is damped so the residual sums stay O(1).
    `blocks` / `block_type` / `base_width` override the depth table (small test networks)."""
    g = torch.Generator().manual_seed(seed)
    bt, nb = RESNET_DEFS[depth] if depth in RESNET_DEFS else (block_type, blocks)
    if blocks is not None:
        nb = blocks
    if block_type is not None:
        bt = block_type
    exp = 4 if bt == "bottleneck" else 1

    def conv(cout, cin, k, damp=1.0):
        w = torch.randn(cout, cin, k, k, generator=g) * ((2.0 / (cin * k * k)) ** 0.5 * damp)
        b = torch.randn(cout, generator=g) * 0.01
        return w, b

    R = {"block_type": bt}
    R["conv1_w"], R["conv1_b"] = conv(base_width, 3, 7)
    layers = []
    cin = base_width
    for li, n in enumerate(nb):
        width = base_width * (2 ** li)
        stride = 1 if li == 0 else 2
        layer = []
        for bi in range(n):
            st = stride if bi == 0 else 1
            cout = width * exp
            if bt == "bottleneck":
                w1, b1 = conv(width, cin, 1)
                w2, b2 = conv(width, width, 3)
                w3, b3 = conv(cout, width, 1, damp=0.3)
                convs = [(w1, b1, 1, 0), (w2, b2, st, 1), (w3, b3, 1, 0)]
            else:
                w1, b1 = conv(width, cin, 3)
                w2, b2 = conv(cout, width, 3, damp=0.3)
                convs = [(w1, b1, st, 1), (w2, b2, 1, 1)]
            sc = None
            if st != 1 or cin != cout:
                ws, bs = conv(cout, cin, 1, damp=0.7)
                sc = (ws, bs, st)
            layer.append(dict(convs=convs, shortcut=sc))
            cin = cout
        layers.append(layer)
    R["trunk_blocks"] = [b for layer in layers[:3] for b in layer]
    R["head_blocks"] = list(layers[3])
    R["cls_w"] = torch.randn(n_classes, cin, generator=g) * 0.01
    R["cls_b"] = torch.zeros(n_classes)
    R["bbox_w"] = torch.randn(4 * n_classes, cin, generator=g) * 0.001
    R["bbox_b"] = torch.zeros(4 * n_classes)
    R["bbox_mean"], R["bbox_std"] = ([0.0, 0.0, 0.0, 0.0], [0.1, 0.1, 0.2, 0.2]) if bbox_norm else (None, None)
    return R


RESNET_MPN_REGIONS = [0, 1, 2, 3, 1]  # four Foveal classification towers + the box tower (as multipathnet.lua:78-111's five towers)


def synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=557, regions=RESNET_MPN_REGIONS, **kw):
    """MultiPathNet on a ResNet backbone (BASELINE configs[3]; this library's extension — the reference has no such model):
    shared trunk, one layer4 copy per Foveal tower, K integral classifiers on the concatenated classification towers, the
    last tower feeding the box regressor."""
    R = synthetic_resnet_params(depth=depth, n_classes=n_classes, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 7)
    base = R["head_blocks"]
    towers = []
    for t, _ in enumerate(regions):
        if t == 0:
            towers.append(base)
            continue
        tw = []
        for b in base:  # same shapes, fresh weights
            convs = [(torch.randn(w.shape, generator=g) * w.std(), torch.randn(bb.shape, generator=g) * 0.01, st, pd) for (w, bb, st, pd) in b["convs"]]
            sc = None if b["shortcut"] is None else (torch.randn(b["shortcut"][0].shape, generator=g) * b["shortcut"][0].std(),
                                                     torch.randn(b["shortcut"][1].shape, generator=g) * 0.01, b["shortcut"][2])
            tw.append(dict(convs=convs, shortcut=sc))
        towers.append(tw)
    out_c = R["cls_w"].shape[1]
    R["head_towers"], R["head_regions"] = towers, list(regions)
    nf = len(regions) - 1
    R["cls_w"] = torch.randn(n_integral * n_classes, nf * out_c, generator=g) * 0.01
    R["cls_b"] = torch.zeros(n_integral * n_classes)
    R["bbox_w"] = torch.randn(4 * n_classes, out_c, generator=g) * 0.001
    R["bbox_b"] = torch.zeros(4 * n_classes)
    R["n_integral"], R["n_classes"] = n_integral, n_classes
    return R


def resnet_params_numpy(R):
    """the same parameters in the oracle's (numpy) form"""
    n = lambda t: t.detach().cpu().numpy()
    blk = lambda b: dict(convs=[(n(w), n(bb), st, pd) for (w, bb, st, pd) in b["convs"]],
                         shortcut=None if b["shortcut"] is None else (n(b["shortcut"][0]), n(b["shortcut"][1]), b["shortcut"][2]))
    out = {k: n(R[k]) for k in ("conv1_w", "conv1_b", "cls_w", "cls_b", "bbox_w", "bbox_b")}
    out["trunk_blocks"] = [blk(b) for b in R["trunk_blocks"]]
    out["head_blocks"] = [blk(b) for b in R["head_blocks"]]
    if "head_towers" in R:
        out["head_towers"] = [[blk(b) for b in tw] for tw in R["head_towers"]]
        out["head_regions"], out["n_integral"], out["n_classes"] = R["head_regions"], R["n_integral"], R["n_classes"]
    out["bbox_mean"], out["bbox_std"] = R["bbox_mean"], R["bbox_std"]
    return out


def ResNetFRCNN(params, **kw):
    """models/resnet.lua graph as one device pipeline: pooled 14x14, ImagenetTransformer (resnet.lua:46,52)"""
    kw.setdefault("pooled", 14)
    kw.setdefault("transformer", IMAGENET_TRANSFORMER)
    return FastRCNN(params, **kw)


# ---- Inception-v3 (models/inceptionv3.lua:27-43; the public Inception-v3 definition, BN folded) as two op lists ------------------
class _GraphBuilder(object):
    """builds an op list over numbered tensors; kinds: 0 conv(+ReLU), 1 max-pool, 2 average pool (count_include_pad)"""

    def __init__(self, c0, gen, width=1.0):
        self.ops, self.tc, self.g, self.width = [], [c0], gen, width

    def ch(self, c):  # channel counts scale with `width`, kept multiples of 16 (test-size networks)
        return c if self.width == 1.0 else max(16, int(round(c * self.width / 16.0)) * 16)

    def tensor(self, c):
        self.tc.append(c)
        return len(self.tc) - 1

    def conv(self, src, cout, k, s=1, p=0, dst=None, off=0, damp=1.0):
        kh, kw = (k, k) if isinstance(k, int) else k
        ph, pw = (p, p) if isinstance(p, int) else p
        cin = self.tc[src]
        if dst is None:
            dst = self.tensor(cout)
        w = torch.randn(cout, cin, kh, kw, generator=self.g) * ((2.0 / (cin * kh * kw)) ** 0.5 * damp)
        b = torch.randn(cout, generator=self.g) * 0.01
        self.ops.append(dict(kind=0, src=src, dst=dst, off=off, cin=cin, cout=cout, kh=kh, kw=kw, sh=s, sw=s, ph=ph, pw=pw, relu=1, w=w, b=b))
        return dst

    def pool(self, kind, src, k, s, p, dst=None, off=0, ceil=0):
        c = self.tc[src]
        if dst is None:
            dst = self.tensor(c)
        self.ops.append(dict(kind=kind, src=src, dst=dst, off=off, cin=c, cout=c, kh=k, kw=k, sh=s, sw=s, ph=p, pw=p, relu=0, w=None, b=None, ceil=ceil))
        return dst

    def grouped_conv(self, src, cout, k, groups, s=1, p=0, scale=1.0):
        """cudnn.SpatialConvolution(..., groups): one op per group — it reads its channel range of `src` (src_off) and writes its range
        of the output (off)"""
        cin = self.tc[src]
        assert cin % (8 * groups) == 0 and cout % (8 * groups) == 0, "group channel ranges must be whole 8-channel blocks"
        cg, og = cin // groups, cout // groups
        dst = self.tensor(cout)
        for gi in range(groups):
            w = torch.randn(og, cg, k, k, generator=self.g) * ((2.0 / (cg * k * k)) ** 0.5 * scale)
            b = torch.randn(og, generator=self.g) * 0.01
            self.ops.append(dict(kind=0, src=src, dst=dst, off=gi * og, src_off=gi * cg, cin=cg, cout=og, kh=k, kw=k, sh=s, sw=s, ph=p, pw=p, relu=1, w=w, b=b))
        return dst

    def lrn(self, src, size=5, alpha=1e-4, beta=0.75, k=1.0):
        """nn.SpatialCrossMapLRN(size, alpha, beta, k)"""
        c = self.tc[src]
        dst = self.tensor(c)
        self.ops.append(dict(kind=3, src=src, dst=dst, off=0, cin=c, cout=c, kh=size, kw=1, sh=1, sw=1, ph=0, pw=0, relu=0, w=None, b=None,
                             lrn=(alpha, beta, k)))
        return dst

    # Inception modules: every branch's last op writes into its slice of the module's output tensor (DepthConcat)
    def inception_a(self, x, pf):
        c = self.ch
        out = self.tensor(c(64) + c(64) + c(96) + c(pf)); o = 0
        self.conv(x, c(64), 1, dst=out, off=o); o += c(64)
        t = self.conv(x, c(48), 1); self.conv(t, c(64), 5, p=2, dst=out, off=o); o += c(64)
        t = self.conv(x, c(64), 1); t = self.conv(t, c(96), 3, p=1); self.conv(t, c(96), 3, p=1, dst=out, off=o); o += c(96)
        t = self.pool(2, x, 3, 1, 1); self.conv(t, c(pf), 1, dst=out, off=o)
        return out

    def inception_b(self, x):
        c = self.ch
        cin = self.tc[x]
        out = self.tensor(c(384) + c(96) + cin); o = 0
        self.conv(x, c(384), 3, s=2, dst=out, off=o); o += c(384)
        t = self.conv(x, c(64), 1); t = self.conv(t, c(96), 3, p=1); self.conv(t, c(96), 3, s=2, dst=out, off=o); o += c(96)
        self.pool(1, x, 3, 2, 0, dst=out, off=o)
        return out

    def inception_c(self, x, c7):
        c = self.ch
        c7 = c(c7)
        out = self.tensor(4 * c(192)); o = 0
        self.conv(x, c(192), 1, dst=out, off=o); o += c(192)
        t = self.conv(x, c7, 1); t = self.conv(t, c7, (1, 7), p=(0, 3)); self.conv(t, c(192), (7, 1), p=(3, 0), dst=out, off=o); o += c(192)
        t = self.conv(x, c7, 1); t = self.conv(t, c7, (7, 1), p=(3, 0)); t = self.conv(t, c7, (1, 7), p=(0, 3))
        t = self.conv(t, c7, (7, 1), p=(3, 0)); self.conv(t, c(192), (1, 7), p=(0, 3), dst=out, off=o); o += c(192)
        t = self.pool(2, x, 3, 1, 1); self.conv(t, c(192), 1, dst=out, off=o)
        return out

    def inception_d(self, x):
        c = self.ch
        cin = self.tc[x]
        out = self.tensor(c(320) + c(192) + cin); o = 0
        t = self.conv(x, c(192), 1); self.conv(t, c(320), 3, s=2, dst=out, off=o); o += c(320)
        t = self.conv(x, c(192), 1); t = self.conv(t, c(192), (1, 7), p=(0, 3)); t = self.conv(t, c(192), (7, 1), p=(3, 0))
        self.conv(t, c(192), 3, s=2, dst=out, off=o); o += c(192)
        self.pool(1, x, 3, 2, 0, dst=out, off=o)
        return out

    def inception_e(self, x):
        c = self.ch
        out = self.tensor(c(320) + 4 * c(384) + c(192)); o = 0
        self.conv(x, c(320), 1, dst=out, off=o, damp=0.7); o += c(320)
        t = self.conv(x, c(384), 1)
        self.conv(t, c(384), (1, 3), p=(0, 1), dst=out, off=o); o += c(384)
        self.conv(t, c(384), (3, 1), p=(1, 0), dst=out, off=o); o += c(384)
        t = self.conv(x, c(448), 1); t = self.conv(t, c(384), 3, p=1)
        self.conv(t, c(384), (1, 3), p=(0, 1), dst=out, off=o); o += c(384)
        self.conv(t, c(384), (3, 1), p=(1, 0), dst=out, off=o); o += c(384)
        t = self.pool(2, x, 3, 1, 1); self.conv(t, c(192), 1, dst=out, off=o)
        return out


def synthetic_inception_v3_params(n_classes=21, seed=557, width=1.0, bbox_norm=True):
    """Inception-v3 Fast R-CNN (inceptionv3.lua:27-43) as op lists: trunk = stem + Mixed_5b..6e (net:get(1..25), the 17x17 stage
    at stride 299/17), head = Mixed_7a..7c + the final average pool (net:get(26..30)); ROIPooling(17,17) at 17/299.
    BN folded, He-scaled random weights; `width` < 1 scales every channel count (test-size networks).  The reference's `.t7`
    is absent: this is the public Inception-v3 structure (PARITY UNPINNED)."""
    g = torch.Generator().manual_seed(seed)
    tb = _GraphBuilder(3, g, width)
    c = tb.ch
    x = tb.conv(0, c(32), 3, s=2)
    x = tb.conv(x, c(32), 3)
    x = tb.conv(x, c(64), 3, p=1)
    x = tb.pool(1, x, 3, 2, 0)
    x = tb.conv(x, c(80), 1)
    x = tb.conv(x, c(192), 3)
    x = tb.pool(1, x, 3, 2, 0)
    x = tb.inception_a(x, 32)
    x = tb.inception_a(x, 64)
    x = tb.inception_a(x, 64)
    x = tb.inception_b(x)
    for c7 in (128, 160, 160, 192):
        x = tb.inception_c(x, c7)
    feat = x
    hb = _GraphBuilder(tb.tc[feat], g, width)
    y = hb.inception_d(0)
    y = hb.inception_e(y)
    y = hb.inception_e(y)
    out_c = hb.tc[y]
    G = dict(trunk_ops=tb.ops, trunk_tensor_c=tb.tc, feat_tensor=feat, head_ops=hb.ops, head_tensor_c=hb.tc, out_tensor=y)
    G["cls_w"] = torch.randn(n_classes, out_c, generator=g) * 0.01
    G["cls_b"] = torch.zeros(n_classes)
    G["bbox_w"] = torch.randn(4 * n_classes, out_c, generator=g) * 0.001
    G["bbox_b"] = torch.zeros(4 * n_classes)
    G["bbox_mean"], G["bbox_std"] = ([0.0, 0.0, 0.0, 0.0], [0.1, 0.1, 0.2, 0.2]) if bbox_norm else (None, None)
    return G


def synthetic_alexnet_params(n_classes=21, seed=557, width=1.0, fc_dim=4096, bbox_norm=True):
    """BASELINE configs[0]: AlexNet / CaffeNet Fast R-CNN (models/alexnet.lua:14-27) as op lists.  `features` = the public Fast R-CNN
    CaffeNet trunk — conv1 11x11/4 pad 5 (96) - ReLU - max-pool 3x3/2 pad 1 (ceil) - LRN(5, 1e-4, 0.75) - conv2 5x5 pad 2 groups 2 (256) -
    ReLU - max-pool - LRN - conv3 3x3 (384) - conv4 3x3 groups 2 (384) - conv5 3x3 groups 2 (256), stride 16 — then
    inn.ROIPooling(6,6,1/16), and `top` = fc6 / fc7 (4096, Dropout = identity in evaluate mode) expressed on the pooled [256,6,6] map as
    a 6x6 and a 1x1 convolution (the View(-1) flattening is exactly the [cout][cin][6][6] weight order).  The reference's
    imagenet_pretrained_alexnet.t7 is absent: structure from the public definition, seeded weights (PARITY UNPINNED).
    `width` < 1 scales the channel counts (test-size networks; multiples of 16 so that the two groups stay whole channel blocks)."""
    g = torch.Generator().manual_seed(seed)
    tb = _GraphBuilder(3, g, width)
    c = tb.ch
    x = tb.conv(0, c(96), 11, s=4, p=5, damp=1.0 / 70.0)   # sees mean-subtracted 0..255 pixels (as synthetic_params' first conv)
    x = tb.pool(1, x, 3, 2, 1, ceil=1)
    x = tb.lrn(x)
    x = tb.grouped_conv(x, c(256), 5, 2, p=2)
    x = tb.pool(1, x, 3, 2, 1, ceil=1)
    x = tb.lrn(x)
    x = tb.conv(x, c(384), 3, p=1)
    x = tb.grouped_conv(x, c(384), 3, 2, p=1)
    x = tb.grouped_conv(x, c(256), 3, 2, p=1)
    feat = x
    hb = _GraphBuilder(tb.tc[feat], g, 1.0)
    y = hb.conv(0, fc_dim, 6)     # fc6 on the 6x6 pooled map
    y = hb.conv(y, fc_dim, 1)     # fc7
    G = dict(trunk_ops=tb.ops, trunk_tensor_c=tb.tc, feat_tensor=feat, head_ops=hb.ops, head_tensor_c=hb.tc, out_tensor=y)
    G["cls_w"] = torch.randn(n_classes, fc_dim, generator=g) * 0.01
    G["cls_b"] = torch.zeros(n_classes)
    G["bbox_w"] = torch.randn(4 * n_classes, fc_dim, generator=g) * 0.001
    G["bbox_b"] = torch.zeros(4 * n_classes)
    G["bbox_mean"], G["bbox_std"] = ([0.0, 0.0, 0.0, 0.0], [0.1, 0.1, 0.2, 0.2]) if bbox_norm else (None, None)
    return G


def AlexNetFRCNN(params, **kw):
    """models/alexnet.lua graph as one device pipeline: ROIPooling(6,6,1/16), RossTransformer (alexnet.lua:23,32)"""
    kw.setdefault("pooled", 6)
    kw.setdefault("spatial_scale", 1.0 / 16)
    return FastRCNN(params, **kw)


def synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557, regions=RESNET_MPN_REGIONS, **kw):
    """BASELINE configs[4]: MultiPathNet on the Inception-v3 backbone (this library's extension, as synthetic_resnet_mpn_params):
    one Mixed_7a..7c copy per Foveal tower, K integral classifiers, the last tower feeds the box regressor."""
    G = synthetic_inception_v3_params(n_classes=n_classes, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 9)
    towers = [G["head_ops"]]
    for _ in regions[1:]:
        tw = []
        for o in G["head_ops"]:
            if o["kind"] == 0:
                tw.append(dict(o, w=torch.randn(o["w"].shape, generator=g) * o["w"].std(), b=torch.randn(o["b"].shape, generator=g) * 0.01))
            else:
                tw.append(dict(o))
        towers.append(tw)
    out_c = G["bbox_w"].shape[1]
    G["head_towers"], G["head_regions"] = towers, list(regions)
    G["cls_w"] = torch.randn(n_integral * n_classes, (len(regions) - 1) * out_c, generator=g) * 0.01
    G["cls_b"] = torch.zeros(n_integral * n_classes)
    G["bbox_w"] = torch.randn(4 * n_classes, out_c, generator=g) * 0.001
    G["bbox_b"] = torch.zeros(4 * n_classes)
    G["n_integral"], G["n_classes"] = n_integral, n_classes
    return G


INCEPTION_TRANSFORMER = dict(mean=(1.0, 1.0, 1.0), std=None, scale=2.0, swap=(0, 1, 2))  # fbcoco.ImageTransformer({1,1,1},nil,2), inceptionv3.lua:52


def graph_params_numpy(G):
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    cp = lambda ops: [dict(o, w=n(o["w"]), b=n(o["b"])) for o in ops]
    out = dict(G, trunk_ops=cp(G["trunk_ops"]), head_ops=cp(G["head_ops"]))
    if "head_towers" in G:
        out["head_towers"] = [cp(tw) for tw in G["head_towers"]]
    for k in ("cls_w", "cls_b", "bbox_w", "bbox_b"):
        out[k] = n(G[k])
    return out


def InceptionFRCNN(params, **kw):
    """models/inceptionv3.lua graph as one device pipeline: ROIPooling(17,17) at 17/299, ImageTransformer({1,1,1},nil,2)"""
    kw.setdefault("pooled", 17)
    kw.setdefault("spatial_scale", 17.0 / 299.0)
    kw.setdefault("transformer", INCEPTION_TRANSFORMER)
    return FastRCNN(params, **kw)


# models/multipathnet.lua:78-111: four foveal towers (region i; conv4 if i<=3, conv3 if i==1) + the "het" tower
# (region 2, all three maps) whose output feeds the box regressor.  Regions are 0-based here.
MPN_TOWERS = [dict(region=0, use4=1, use3=1), dict(region=1, use4=1, use3=0), dict(region=2, use4=1, use3=0),
              dict(region=3, use4=0, use3=0), dict(region=1, use4=1, use3=1)]


def conv_tap_indices(cfg):
    """0-based conv-layer indices of the last conv before the 3rd and 4th pools (VGG-16: conv3_3 = 6, conv4_3 = 9)."""
    convs, pools = -1, []
    for item in cfg:
        if item == "P":
            pools.append(convs)
        else:
            convs += 1
    return pools[2], pools[3]


def synthetic_mpnet_params(cfg=VGG16_CFG, pooled=7, fc_dim=4096, n_classes=81, n_integral=6, seed=557, towers=MPN_TOWERS):
    """Seeded weights for the MultiPathNet graph: shared trunk + per-tower {1x1 mix, fc6, fc7} + K integral
    classifiers + one box regressor (multipathnet.lua:64-120, model_utils.lua:275-317)."""
    P = synthetic_params(cfg, pooled, fc_dim, n_classes, seed)
    g = torch.Generator().manual_seed(seed + 1)
    cout, _ = cfg_layers(cfg)
    t3, t4 = conv_tap_indices(cfg)
    c5, c4, c3 = cout[-1], cout[t4], cout[t3]
    P["towers"] = []
    for T in towers:
        tf = c5 + (c4 if T["use4"] else 0) + (c3 if T["use3"] else 0)
        k6 = c5 * pooled * pooled
        P["towers"].append(dict(
            T, total_feat=tf,
            # after Normalize*1000 each map has RMS ~ 1000/sqrt(C*49); scale the mix so its output is O(1)
            mix_w=torch.randn(c5, tf, generator=g) * (2.0 / tf) ** 0.5 * 0.15, mix_b=torch.randn(c5, generator=g) * 0.01,
            fc6_w=torch.randn(fc_dim, k6, generator=g) * (2.0 / k6) ** 0.5, fc6_b=torch.randn(fc_dim, generator=g) * 0.01,
            fc7_w=torch.randn(fc_dim, fc_dim, generator=g) * (2.0 / fc_dim) ** 0.5, fc7_b=torch.randn(fc_dim, generator=g) * 0.01))
    nf = len(towers) - 1
    P["cls_w"] = torch.randn(n_integral * n_classes, nf * fc_dim, generator=g) * 0.01  # K classifier clones stacked
    P["cls_b"] = torch.zeros(n_integral * n_classes)
    P["bbox_w"] = torch.randn(4 * n_classes, fc_dim, generator=g) * 0.001
    P["bbox_b"] = torch.zeros(4 * n_classes)
    P["n_integral"], P["n_classes"] = n_integral, n_classes
    for k in ("fc6_w", "fc6_b", "fc7_w", "fc7_b"):
        P.pop(k, None)
    return P


def MultiPathNet(params, **kw):
    """models/multipathnet.lua graph as one device pipeline (same object type; noSoftMax like model_utils.lua:314)."""
    net = FastRCNN(params, **kw)
    net.noSoftMax = True
    return net


class FastRCNN(object):
    """Trunk + ROI head + post-processing as one device pipeline (models/vgg.lua:23-31 graph)."""

    def __init__(self, params, cfg=VGG16_CFG, pooled=7, spatial_scale=1.0 / 16, transformer=None, max_h=600, max_w=1000,
                 max_rois=1000, nms_thresh=0.3, score_thresh=-1.5, top_k=100, num_iter=1, bbox_voting=False, bbox_vote_thresh=0.5,
                 bbox_vote_score_pow=1.0, scale=None, max_size=None, bf16=False, use_rbox_scores=False, roi_bin_rule=0, fc_arith=None):
        """scale / max_size: getImages' rescaling (ImageDetect.lua:34-43) on the device; None feeds images as they are.
        roi_bin_rule: 0 = inn.ROIPooling's CUDA-branch bins, 1 = its CPU branch (crop + SpatialAdaptiveMaxPooling), include/mpn.h MPN_ROI_BINS_*.
        num_iter / bbox_voting / use_rbox_scores: opt.test_num_iterative_loc / test_bbox_voting / test_use_rbox_scores
        (Tester_FRCNN.lua:82-99,118-124) inside the fused test_one."""
        _lib.require_gpu()
        lib = _lib.load()
        self.is_resnet = "trunk_blocks" in params
        self.is_graph = "trunk_ops" in params
        cout, pool = ([], []) if (self.is_resnet or self.is_graph) else cfg_layers(cfg)
        self.is_mpnet = "towers" in params
        self.n_classes = params["n_classes"] if (self.is_mpnet or "head_towers" in params) else params["cls_w"].shape[0]
        if self.is_resnet or self.is_graph:
            self.fc_dim = params["bbox_w"].shape[1]
        else:
            self.fc_dim = params["towers"][0]["fc7_w"].shape[0] if self.is_mpnet else params["fc7_w"].shape[0]
        self.noSoftMax = False
        self.max_rois = max_rois
        c = FrcnnConfig()
        self._cout = (C.c_int * len(cout))(*cout)
        self._pool = (C.c_int * len(pool))(*pool)
        c.n_conv = len(cout)
        c.conv_cout = C.cast(self._cout, C.POINTER(C.c_int))
        c.pool_after = C.cast(self._pool, C.POINTER(C.c_int))
        c.pooled_h = c.pooled_w = pooled
        c.spatial_scale = spatial_scale
        c.fc_dim = self.fc_dim
        c.n_classes = self.n_classes
        c.max_h, c.max_w, c.max_rois = max_h, max_w, max_rois
        tf = transformer or dict(mean=(102.9801, 115.9465, 122.7717), std=None, scale=255.0, swap=(2, 1, 0))
        c.tf_scale = tf["scale"]
        for i in range(3):
            c.tf_mean[i] = tf["mean"][i]
            c.tf_std[i] = tf["std"][i] if tf["std"] else 0.0
            c.tf_swap[i] = tf["swap"][i]
        bm, bs = params.get("bbox_mean"), params.get("bbox_std")
        for i in range(4):
            c.bbox_mean[i] = bm[i] if bm is not None else 0.0
            c.bbox_std[i] = bs[i] if bs is not None else 0.0
        c.nms_thresh, c.score_thresh, c.top_k = nms_thresh, score_thresh, top_k
        c.num_iter, c.bbox_voting, c.bbox_vote_thresh, c.bbox_vote_score_pow = num_iter, int(bbox_voting), bbox_vote_thresh, bbox_vote_score_pow
        self.num_iter = num_iter
        c.scale_target, c.scale_max = float(scale or 0.0), float(max_size or 0.0)
        c.use_rbox_scores = int(bool(use_rbox_scores))
        c.roi_bin_rule = int(roi_bin_rule)
        # fc_arith: 0 = fc6 on the fp32 matrix pipe (default), 1 / "split3" = the three-plane bf16 split with fp32 accumulation (include/mpn.h
        # MPN_FC_SPLIT3; plain VGG / AlexNet-free pipelines only).  MPN_FC_ARITH=split3 in the environment forces it on for every plain VGG
        # pipeline a process builds (how the fp32 parity suite is run against it: tools/r06_split3_gate.sh).
        if fc_arith is None:
            fc_arith = os.environ.get("MPN_FC_ARITH", "0")
        fc_arith = {"0": 0, "fp32": 0, "1": 1, "split3": 1}[str(fc_arith)]
        c.fc_arith = 0 if (self.is_resnet or self.is_graph) else fc_arith   # the VGG pipelines (Fast R-CNN, MultiPathNet towers) have fc6 / fc7
        self.fc_arith = c.fc_arith
        self.scale, self.max_size = scale, max_size
        self._cfg = c
        dev = torch.device("cuda", torch.cuda.current_device())
        d = lambda t: t.to(dev, torch.float32).contiguous()
        self._h = C.c_void_p()
        if self.is_graph:
            keep = []

            def mk(ops):
                arr = (GraphOp * len(ops))()
                for i, o in enumerate(ops):
                    a = arr[i]
                    a.kind, a.src, a.dst, a.dst_c_off, a.cin, a.cout = o["kind"], o["src"], o["dst"], o["off"], o["cin"], o["cout"]
                    a.kh, a.kw, a.sh, a.sw, a.ph, a.pw, a.relu = o["kh"], o["kw"], o["sh"], o["sw"], o["ph"], o["pw"], o["relu"]
                    a.src_c_off, a.ceil_mode = o.get("src_off", 0), o.get("ceil", 0)
                    if o["kind"] == 3:
                        a.lrn_alpha, a.lrn_beta, a.lrn_k = o["lrn"]
                    if o["kind"] == 0:
                        wd, bd = d(o["w"]), d(o["b"])
                        keep.extend([wd, bd])
                        a.w, a.b = _f(wd), _f(bd)
                return arr

            gw = GraphWeights()
            towers = params.get("head_towers") or [params["head_ops"]]
            self._g_arrays = [mk(params["trunk_ops"]), mk([o for tw in towers for o in tw]), (C.c_int * len(params["trunk_tensor_c"]))(*params["trunk_tensor_c"]),
                              (C.c_int * len(params["head_tensor_c"]))(*params["head_tensor_c"])]
            ga = self._g_arrays
            gw.n_trunk_ops, gw.trunk_ops = len(params["trunk_ops"]), C.cast(ga[0], C.POINTER(GraphOp))
            gw.n_head_ops, gw.head_ops = len(params["head_ops"]), C.cast(ga[1], C.POINTER(GraphOp))
            gw.n_trunk_tensors, gw.trunk_tensor_c = len(params["trunk_tensor_c"]), C.cast(ga[2], C.POINTER(C.c_int))
            gw.n_head_tensors, gw.head_tensor_c = len(params["head_tensor_c"]), C.cast(ga[3], C.POINTER(C.c_int))
            gw.feat_tensor, gw.out_tensor, gw.bf16 = params["feat_tensor"], params["out_tensor"], int(bool(bf16))
            if "head_towers" in params:
                gw.n_heads, gw.n_integral = len(towers), params["n_integral"]
                for t, rg in enumerate(params["head_regions"]):
                    gw.head_region[t] = rg
                self.noSoftMax = True
            heads = [d(params[k]) for k in ("cls_w", "cls_b", "bbox_w", "bbox_b")]
            check(lib.mpn_graph_create(C.byref(c), C.byref(gw), *[_f(t) for t in heads], C.byref(self._h)), "mpn_graph_create")
            torch.cuda.synchronize()
            self._finish_init(lib, dev, top_k)
            return
        if self.is_resnet:
            convs, nconv, hassc = [], [], []  # execution order: conv1, then per block its convolutions (+ shortcut)
            convs.append((params["conv1_w"], params["conv1_b"], 2, 3))
            towers = params.get("head_towers") or [params["head_blocks"]]
            blocks = list(params["trunk_blocks"]) + [b for tw in towers for b in tw]
            for b in blocks:
                convs += list(b["convs"])
                nconv.append(len(b["convs"]))
                hassc.append(0 if b["shortcut"] is None else 1)
                if b["shortcut"] is not None:
                    convs.append((b["shortcut"][0], b["shortcut"][1], b["shortcut"][2], 0))
            keep = [(d(w), d(bb)) for (w, bb, _, _) in convs]
            ia = lambda vals: (C.c_int * len(vals))(*[int(v) for v in vals])
            rw = ResnetWeights()
            rw.n_convs = len(convs)
            self._rw_arrays = [(f32p * len(keep))(*[_f(w) for w, _ in keep]), (f32p * len(keep))(*[_f(bb) for _, bb in keep]),
                               ia([w.shape[1] for w, _ in keep]), ia([w.shape[0] for w, _ in keep]), ia([w.shape[2] for w, _ in keep]),
                               ia([c_[2] for c_ in convs]), ia([c_[3] for c_ in convs]), ia(nconv), ia(hassc)]
            a = self._rw_arrays
            rw.w, rw.b = C.cast(a[0], C.POINTER(f32p)), C.cast(a[1], C.POINTER(f32p))
            rw.cin, rw.cout, rw.ksize, rw.stride, rw.pad = [C.cast(x, C.POINTER(C.c_int)) for x in a[2:7]]
            rw.n_blocks = len(blocks)
            rw.block_n_convs, rw.block_has_shortcut = C.cast(a[7], C.POINTER(C.c_int)), C.cast(a[8], C.POINTER(C.c_int))
            rw.n_trunk_blocks = len(params["trunk_blocks"])
            rw.bf16 = int(bool(bf16))
            if "head_towers" in params:
                rw.n_heads, rw.n_integral = len(towers), params["n_integral"]
                for t, rg in enumerate(params["head_regions"]):
                    rw.head_region[t] = rg
                self.noSoftMax = True
            heads = [d(params[k]) for k in ("cls_w", "cls_b", "bbox_w", "bbox_b")]
            check(lib.mpn_resnet_create(C.byref(c), C.byref(rw), *[_f(t) for t in heads], C.byref(self._h)), "mpn_resnet_create")
            torch.cuda.synchronize()
            self._finish_init(lib, dev, top_k)
            return
        cw = [d(w) for w in params["conv_w"]]
        cb = [d(b) for b in params["conv_b"]]
        wp = (f32p * len(cw))(*[_f(w) for w in cw])
        bp = (f32p * len(cb))(*[_f(b) for b in cb])
        if self.is_mpnet:
            mw = MpnetWeights()
            tow = params["towers"]
            mw.n_towers = len(tow)
            mw.tap_conv3, mw.tap_conv4 = conv_tap_indices(cfg)
            mw.n_integral = params["n_integral"]
            mw.conv345_unnormalized = 0 if params.get("conv345_norm", True) else 1  # opt.model_conv345_norm (model_utils.lua:209)
            keep = []
            for t, T in enumerate(tow):
                mw.region[t], mw.use_conv4[t], mw.use_conv3[t] = T["region"], T["use4"], T["use3"]
                for name in ("mix_w", "mix_b", "fc6_w", "fc6_b", "fc7_w", "fc7_b"):
                    dt = d(T[name])
                    keep.append(dt)
                    getattr(mw, name)[t] = _f(dt)
            heads = [d(params[k]) for k in ("cls_w", "cls_b", "bbox_w", "bbox_b")]
            check(lib.mpn_mpnet_create(C.byref(c), wp, bp, C.byref(mw), *[_f(t) for t in heads], C.byref(self._h)), "mpn_mpnet_create")
        else:
            keep = [d(params[k]) for k in ("fc6_w", "fc6_b", "fc7_w", "fc7_b", "cls_w", "cls_b", "bbox_w", "bbox_b")]
            check(lib.mpn_frcnn_create(C.byref(c), wp, bp, *[_f(t) for t in keep], C.byref(self._h)), "mpn_frcnn_create")
        torch.cuda.synchronize()
        self._finish_init(lib, dev, top_k)

    def _finish_init(self, lib, dev, top_k):
        self._lib = lib
        self.device = dev
        self._dets = torch.zeros((top_k * 4 + 64, 6), dtype=torch.float32, device=dev)
        self._n_dets = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dets2 = [torch.zeros_like(self._dets) for _ in range(2)]
        self._n_dets2 = [torch.zeros_like(self._n_dets) for _ in range(2)]
        self._pipe_seq = 0

    def close(self):
        """mpn_frcnn_destroy now (streams, events, buffers) instead of at garbage collection: a process holds a few HIP hardware queues, and
        the copy / side streams of an idle handle still occupy them"""
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mpn_frcnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect(self, image, boxes, recompute_features=True, clamp=True):
        """image [3,H,W] fp32 in [0,1] (device), boxes [N,4] (device) -> (scores [N,C], boxes [N,4C] decoded).
        clamp=True also applies Tester_FRCNN.lua:75-78's clamp to the image (what testOne does to its first detect());
        clamp=False is ImageDetect:detect's own (unclamped) output.
        recompute_features=False (ImageDetect.lua:107-111) reuses the cached trunk output of the previous call."""
        H, W = image.shape[1:]
        N = boxes.size(0)
        scores = torch.empty((N, self.n_classes), dtype=torch.float32, device=self.device)
        bbox = torch.empty((N, 4 * self.n_classes), dtype=torch.float32, device=self.device)
        img_ptr = _f(image, "image") if recompute_features else None
        check(self._lib.mpn_frcnn_detect(self._h, img_ptr, H, W, _f(boxes, "boxes"), N, _f(scores), _f(bbox), int(bool(clamp)), _stream()),
              "mpn_frcnn_detect")
        return scores, bbox

    def test_one_async(self, image, boxes):
        """Enqueue the whole per-image path; results stay on the device (self._dets / self._n_dets)."""
        H, W = image.shape[1:]
        check(self._lib.mpn_frcnn_test_one(self._h, _f(image, "image"), H, W, _f(boxes, "boxes"), boxes.size(0), _f(self._dets),
                                           self._dets.size(0), _i(self._n_dets), _stream()), "mpn_frcnn_test_one")
        return self._dets, self._n_dets

    def test_one_pipelined(self, image, boxes):
        """Throughput form (mpn_frcnn_test_one_pipelined): returns the (dets, n) buffers this call will fill; they are
        valid after the NEXT call or flush().  Two buffer pairs alternate."""
        H, W = image.shape[1:]
        b = self._pipe_seq & 1
        self._pipe_seq += 1
        dets, n = self._dets2[b], self._n_dets2[b]
        check(self._lib.mpn_frcnn_test_one_pipelined(self._h, _f(image, "image"), H, W, _f(boxes, "boxes"), boxes.size(0), _f(dets),
                                                     dets.size(0), _i(n), _stream()), "mpn_frcnn_test_one_pipelined")
        return dets, n

    def test_one_pipelined_host(self, image_pinned, boxes_pinned):
        """mpn_frcnn_test_one_pipelined_host: the same throughput form fed from (pinned) HOST tensors — the upload runs on the
        handle's copy stream and overlaps the previous image's kernels.  Alternate two host buffers between calls; the call blocks
        the host while the device is more than three images behind."""
        H, W = image_pinned.shape[1:]
        assert not image_pinned.is_cuda and not boxes_pinned.is_cuda and image_pinned.dtype == torch.float32
        b = self._pipe_seq & 1
        self._pipe_seq += 1
        dets, n = self._dets2[b], self._n_dets2[b]
        check(self._lib.mpn_frcnn_test_one_pipelined_host(self._h, C.cast(image_pinned.data_ptr(), f32p), H, W,
                                                          C.cast(boxes_pinned.data_ptr(), f32p), boxes_pinned.size(0), _f(dets),
                                                          dets.size(0), _i(n), _stream()), "mpn_frcnn_test_one_pipelined_host")
        return dets, n

    def flush(self):
        check(self._lib.mpn_frcnn_flush(self._h, _stream()), "mpn_frcnn_flush")

    # ---- proposal (ROI) sharding of one image across ranks: the latency mode (mpn_frcnn_shard_*, include/mpn.h) ----
    def shard_record_floats(self, N, world):
        """(floats of one rank's row record, floats of one rank's class record) for N proposals over `world` ranks."""
        return (int(self._lib.mpn_frcnn_shard_rows_floats(self._h, int(N), int(world))),
                int(self._lib.mpn_frcnn_shard_class_floats(self._h, int(N), int(world))))

    def shard_head(self, image, boxes, rank, world, out=None):
        """Trunk + ROI head on this rank's slice of `boxes` (the WHOLE table is passed) -> its row record."""
        H, W = image.shape[1:]
        N = boxes.size(0)
        if out is None:
            out = torch.empty(self.shard_record_floats(N, world)[0], dtype=torch.float32, device=self.device)
        check(self._lib.mpn_frcnn_shard_head(self._h, _f(image, "image"), H, W, _f(boxes, "boxes"), N, int(rank), int(world), _f(out), _stream()),
              "mpn_frcnn_shard_head")
        return out

    def shard_nms(self, rows_all, N, rank, world, out=None):
        """All ranks' row records [world, rows_floats] -> this rank's class record (select + NMS (+ vote) of its classes)."""
        if out is None:
            out = torch.empty(self.shard_record_floats(N, world)[1], dtype=torch.float32, device=self.device)
        check(self._lib.mpn_frcnn_shard_nms(self._h, _f(rows_all), int(N), int(rank), int(world), _f(out), _stream()), "mpn_frcnn_shard_nms")
        return out

    def shard_finish(self, class_all, N, world):
        """All ranks' class records [world, class_floats] -> (dets, n) as test_one_async; nms_results() holds every class."""
        check(self._lib.mpn_frcnn_shard_finish(self._h, _f(class_all), int(N), int(world), _f(self._dets), self._dets.size(0), _i(self._n_dets),
                                               _stream()), "mpn_frcnn_shard_finish")
        return self._dets, self._n_dets

    def test_one_sharded(self, comm, image, boxes):
        """mpn_frcnn_test_one_sharded over a parallel.Comm: every rank passes the same image and boxes and receives the
        same (dets, n) — bit-identical to test_one_async on one GPU."""
        H, W = image.shape[1:]
        check(self._lib.mpn_frcnn_test_one_sharded(self._h, comm._h, _f(image, "image"), H, W, _f(boxes, "boxes"), boxes.size(0),
                                                   _f(self._dets), self._dets.size(0), _i(self._n_dets), _stream()), "mpn_frcnn_test_one_sharded")
        return self._dets, self._n_dets

    def nms_results(self):
        """Per-class NMS output of the last test_one: (keep [C-1,N,5], keep_idx [C-1,N], n_keep [C-1])."""
        kp, ip, np_ = f32p(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        ms = C.c_int()
        check(self._lib.mpn_frcnn_nms_results(self._h, C.byref(kp), C.byref(ip), C.byref(np_), C.byref(ms)), "nms_results")
        torch.cuda.synchronize()
        ncls, M = self.n_classes - 1, ms.value  # M = rows per class (N * num_iter)
        keep = torch.empty((ncls, M, 5), dtype=torch.float32, device=self.device)
        idx = torch.empty((ncls, M), dtype=torch.int32, device=self.device)
        n = torch.empty(ncls, dtype=torch.int32, device=self.device)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy(C.c_void_p(keep.data_ptr()), kp, C.c_size_t(keep.numel() * 4), 3)
        hip.hipMemcpy(C.c_void_p(idx.data_ptr()), ip, C.c_size_t(idx.numel() * 4), 3)
        hip.hipMemcpy(C.c_void_p(n.data_ptr()), np_, C.c_size_t(n.numel() * 4), 3)
        return keep, idx, n

    def set_graphs(self, on):
        """captured launch graphs (mpn_frcnn_set_graphs): replay the per-image kernel chains with hipGraphLaunch.
        Default OFF (opt-in: MPN_GRAPHS=1 in the environment or set_graphs(True); include/mpn.h)"""
        check(self._lib.mpn_frcnn_set_graphs(self._h, int(bool(on))), "set_graphs")

    def graph_stats(self):
        """(captures, replays) of this handle's launch graphs"""
        c, r = C.c_long(), C.c_long()
        check(self._lib.mpn_frcnn_graph_stats(self._h, C.byref(c), C.byref(r)), "graph_stats")
        return int(c.value), int(r.value)

    PROF_TAGS = ["transform", "conv_wino", "conv_direct", "pool", "roi_pool", "fc6", "fc7", "heads", "post", "select", "nms", "topk"]

    def set_profiling(self, on):
        """HIP-event timing of every kernel group, recorded on the launch stream (include/mpn.h MPN_PROF_*)."""
        check(self._lib.mpn_frcnn_set_profiling(self._h, int(bool(on))), "set_profiling")

    def get_profile(self, reset=True):
        n = len(self.PROF_TAGS)
        ms, cnt = (C.c_double * n)(), (C.c_long * n)()
        check(self._lib.mpn_frcnn_get_profile(self._h, ms, cnt, n, int(reset)), "get_profile")
        return {t: (ms[i], cnt[i]) for i, t in enumerate(self.PROF_TAGS)}

    def debug_tensor(self, name, shape):
        p, n = f32p(), C.c_size_t()
        check(self._lib.mpn_frcnn_debug_tensor(self._h, name.encode(), C.byref(p), C.byref(n)), "debug_tensor")
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        import ctypes
        ctypes.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(out.data_ptr()), p, C.c_size_t(n.value * 4), 3)
        torch.cuda.synchronize()
        return out.view(*shape)
