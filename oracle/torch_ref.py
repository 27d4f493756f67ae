"""PyTorch-CPU (oneDNN) transcriptions of the graph models — TEST INFRASTRUCTURE like the rest of oracle/ (imported by tests/ and by
bench.py's cpu_baseline leg only, never by the product): the C-ORACLE-INDEPENDENT leg of the full-size parity tests: the dense arithmetic (conv2d / max_pool2d / avg_pool2d / linear, the cudnn / nn semantics the reference relies on) comes from
PyTorch, not from oracle/mpn_oracle.c.  Only the ROI pooling's integer binning + max (no arithmetic) is taken from the oracle by the
callers.  Model structure follows models/resnet.lua:28-50 (fb.resnet.torch blocks, BN folded) and models/inceptionv3.lua:27-43 (the
op lists multipathnet_amd.models builds).

bf16=True emulates the device's bf16 graphs: weights and activations rounded to bf16 (round-to-nearest-even), fp32 accumulation,
bias / residual / ReLU in fp32, ONE rounding at every layer output; the head's average pool and the cls / bbox layers stay fp32."""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F


@contextlib.contextmanager
def threads(n=32):
    """oneDNN on one image does not scale past a few dozen threads (bench.py's cpu_baseline found 16 best on the 256-core GPU box)"""
    old = torch.get_num_threads()
    torch.set_num_threads(min(n, max(1, old)))
    try:
        yield
    finally:
        torch.set_num_threads(old)


def _r(x, bf16):
    return x.bfloat16().float() if bf16 else x


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


def conv(x, w, b, stride, padding, relu, residual=None, bf16=False):
    y = F.conv2d(x, _r(_t(w), bf16), None if b is None else _t(b), stride=stride, padding=padding)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return _r(y, bf16)


def resnet_block(x, blk, bf16=False):
    sc = x if blk["shortcut"] is None else conv(x, blk["shortcut"][0], blk["shortcut"][1], blk["shortcut"][2], 0, False, None, bf16)
    y = x
    n = len(blk["convs"])
    for i, (w, b, st, pd) in enumerate(blk["convs"]):
        y = conv(y, w, b, st, pd, True, sc if i == n - 1 else None, bf16)
    return y


def resnet_trunk(x, R, bf16=False):
    """x: the transformed image [3,H,W] (numpy) -> layer3 output [C,H/16,W/16] (numpy)"""
    with torch.no_grad():
        t = _r(_t(x)[None], bf16)
        t = conv(t, R["conv1_w"], R["conv1_b"], 2, 3, True, None, bf16)
        t = F.max_pool2d(t, 3, 2, 1)
        for blk in R["trunk_blocks"]:
            t = resnet_block(t, blk, bf16)
        return t[0].numpy()


def resnet_tower(pooled, blocks, bf16=False):
    """pooled [n,C,14,14] (numpy) -> layer4 -> 7x7 average pool -> [n,F] tensor (fp32)"""
    with torch.no_grad():
        y = _t(pooled)
        for blk in blocks:
            y = resnet_block(y, blk, bf16)
        return y.mean((2, 3))


def graph_run(x0, ops, tensor_c, bf16=False):
    """the op lists of multipathnet_amd.models (include/mpn.h mpn_graph_op) on x0 [B,C0,H,W] tensor; returns the tensor list"""
    ts = [None] * len(tensor_c)
    ts[0] = x0
    with torch.no_grad():
        for o in ops:
            x = ts[o["src"]]
            c0 = o.get("src_off", 0)
            if c0 or o["cin"] != x.shape[1]:
                x = x[:, c0:c0 + o["cin"]]
            if o["kind"] == 0:
                y = F.conv2d(x, _r(_t(o["w"]), bf16), None if o["b"] is None else _t(o["b"]), stride=(o["sh"], o["sw"]), padding=(o["ph"], o["pw"]))
                if o["relu"]:
                    y = F.relu(y)
                y = _r(y, bf16)
            elif o["kind"] == 1:
                y = F.max_pool2d(x, o["kh"], o["sh"], o["ph"], ceil_mode=bool(o.get("ceil", 0)))
            elif o["kind"] == 3:
                y = F.local_response_norm(x, o["kh"], alpha=o["lrn"][0], beta=o["lrn"][1], k=o["lrn"][2])
            else:
                y = _r(F.avg_pool2d(x, o["kh"], o["sh"], o["ph"], count_include_pad=True), bf16)
            d = o["dst"]
            if ts[d] is None:
                ts[d] = torch.zeros((y.shape[0], tensor_c[d]) + tuple(y.shape[2:]), dtype=torch.float32)
            ts[d][:, o["off"]:o["off"] + y.shape[1]] = y
    return ts


def graph_trunk(x, G, bf16=False):
    """x: the transformed image [3,H,W] (numpy) -> the feature tensor [C,h,w] (numpy)"""
    t = _r(_t(x)[None], bf16)
    return graph_run(t, G["trunk_ops"], G["trunk_tensor_c"], bf16)[G["feat_tensor"]][0].numpy()


def graph_tower(pooled, ops, G, bf16=False):
    """pooled [n,C,17,17] (numpy) -> head op list -> global average pool -> [n,F] tensor (fp32)"""
    y = graph_run(_t(pooled), ops, G["head_tensor_c"], bf16)[G["out_tensor"]]
    return y.mean((2, 3))


def heads(f, P, C):
    """cls / bbox layers (fp32) + BBoxNorm on features f [n,F]: (logits [n, C or K*C], deltas [n, 4C])"""
    with torch.no_grad():
        logits = F.linear(f, _t(P["cls_w"]), _t(P["cls_b"])).numpy()
        deltas = F.linear(f, _t(P["bbox_w"]), _t(P["bbox_b"])).numpy()
    if P.get("bbox_mean") is not None:
        deltas = deltas * np.tile(np.asarray(P["bbox_std"], np.float32), C) + np.tile(np.asarray(P["bbox_mean"], np.float32), C)
    return logits, deltas


def vgg_trunk(x, P, cfg, taps=None):
    """VGG `features` (models/vgg.lua:14-27; ceil-mode pools): x the transformed image [3,H,W] (numpy) -> conv5 [C,h,w] (numpy);
    taps (optional dict): filled with the conv3 / conv4 / conv5 maps MultiPathNet pools from (multipathnet.lua:34-46)"""
    from multipathnet_amd import models
    t3, t4 = models.conv_tap_indices(cfg)
    with torch.no_grad():
        t = _t(x)[None]
        li = 0
        for item in cfg:
            if item == "P":
                t = F.max_pool2d(t, 2, 2, ceil_mode=True)
            else:
                t = F.relu(F.conv2d(t, P["conv_w"][li], P["conv_b"][li], padding=1))
                if taps is not None and li == t3:
                    taps["conv3"] = t[0].numpy()
                if taps is not None and li == t4:
                    taps["conv4"] = t[0].numpy()
                li += 1
        if taps is not None:
            taps["conv5"] = t[0].numpy()
        return t[0].numpy()


def mpnet_tower(pools, T, normalized=True):
    """conv345Combine + fc6 / fc7 of one tower (model_utils.lua:209-251, multipathnet.lua:78-111) on the tower's ROI-pooled maps
    `pools` = [conv5 pool, conv4 pool or None, conv3 pool or None] ([n,C,7,7] numpy each) -> fc7 [n,F] tensor"""
    factors = (1.0, 1.0 / 30, 1.0 / 200)
    with torch.no_grad():
        parts = []
        for m, p in enumerate(pools):
            if p is None:
                continue
            x = _t(p)
            if normalized:
                flat = x.reshape(x.shape[0], -1)
                x = (flat / torch.sqrt((flat * flat).sum(1, keepdim=True) + 1e-10)).reshape(x.shape)
            else:
                x = x * np.float32(factors[m])
            parts.append(x)
        x = torch.cat(parts, 1)
        if normalized:
            x = x * 1000.0
        y = F.conv2d(x, T["mix_w"][:, :, None, None], T["mix_b"])
        h = F.relu(F.linear(y.reshape(y.shape[0], -1), T["fc6_w"], T["fc6_b"]))
        return F.relu(F.linear(h, T["fc7_w"], T["fc7_b"]))
