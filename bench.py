#!/usr/bin/env python
"""bench.py — proposals/sec of the per-image detection hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU; launched
by torch.distributed.run directly it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

A step = one pass of the hot path over one synthetic 600x1000 image with 1000 ROIs through VGG-16 Fast R-CNN
(BASELINE configs[1]), starting — as Tester_FRCNN.lua:64-66 / ImageDetect.lua:148-151 do — from a HOST image and HOST
boxes: H2D upload (pinned buffers, the pipeline's copy stream, three staging sets so that it overlaps the previous image's
kernels) -> image transform -> 13 conv / 4 pool trunk -> ROI pool -> fc6/fc7 -> cls/bbox heads -> softmax / BBoxNorm /
decode / clamp -> per-class NMS -> top-100 -> (N>1) RCCL all-gather of the scored-box record through the C ABI
(mpn_gather_dets).  Weights are seeded random (no pretrained blobs offline).  Images shard across ranks (weak scaling, one
image per rank per step); value = all ranks' proposals / max-over-ranks time.  `value_inputs_resident` repeats the
measurement with the image and boxes already in HBM (no per-step upload).

`--mode latency` is the ROI-sharded form of north_star's "images+proposals shard across the 8 GPUs": every rank is handed the SAME
image and proposal table, runs the trunk, the ROI head on its 1/N of the proposals and the NMS on its 1/N of the classes
(mpn_frcnn_test_one_sharded: two RCCL all-gathers of scored boxes); the line it prints has its own metric string (per-image
latency, strong scaling) and is never the headline.

N > 1 details: the launcher's process group is `gloo` (rendezvous, barriers and the MAX over ranks of the elapsed time — CPU
side); the ONLY RCCL communicator a rank holds is the C ABI's mpn_comm.  Ranks pin themselves to the CPUs of their GPU's NUMA
node; each rank rotates four different pinned images (offset by rank), so uploads are not served from a warm cache.
MPN_BENCH_SHARE_GPU=1 (test only, stated in config.parallelism) maps every rank to device 0 and sends the record gather through
gloo, so that `--gpus 2` exercises the launch / rank / drift-stream / reduction logic on a one-GPU box (RCCL refuses two ranks on
one device).

After the timed K steps a `sustained` leg repeats the same host-fed step for >= 3 s (>= 800 steps) with the GPU's clock and power
sampled beside it (sysfs hwmon of the device's PCI function, else rocm-smi): SURVEY §8(d) says "steady state", and a 70 ms timed
region cannot show what DVFS does over a dataset-length run (Tester_FRCNN.lua:150-157).  `--sustained-seconds 0` skips it.

Prints ONE JSON line (rank 0).  Extra objects:
  "roofline"      dominant kernel group (the Winograd convolutions), fp32-MFMA bound.  `achieved` / `frac` count the FLOPs the
                  matrix pipe EXECUTES (Winograd F(2x2,3x3): 16 multiplies per 4 outputs instead of 36 => direct-conv FLOPs /
                  2.25), so frac <= 1; `algorithmic_equiv_*` is the direct-convolution-equivalent rate (can exceed the peak).
                  Per-launch durations come from HIP events recorded on the launch stream in a second, equally long pass.
  "cpu_baseline"  the same path on this box's host cores: PyTorch-CPU (oneDNN conv2d / max_pool2d(ceil_mode) / linear) +
                  the oracle's ROI pool + the reference's own nms.c, at all cores and at 1 thread, plus the plain C oracle
                  port; rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import gc
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is deliberately NOT set here.  A handle drives up to five streams and ROCm maps a process's streams onto four hardware queues;
# 8 queues take 0.07 ms off the host-fed configs[2] line — and cost a SECOND handle in the same process 20-25 %: the default line's auxiliary
# split3 leg 2.90 -> 3.54-3.71 ms.  profiles/r06_hw_queues.txt)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (already exported on the GPU boxes: RCCL between processes needs the dmabuf IPC path)

H, W, N_ROIS, N_CLASSES = 600, 1000, 1000, 21
FP32_MFMA_PEAK = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
WINO_MUL_RATIO = 36.0 / 16.0  # direct 3x3 multiplies per Winograd F(2x2,3x3) multiply


def conv_flops(cfg, h, w):
    """per-layer algorithmic FLOPs (2*H*W*Cin*9*Cout with the REAL Cin) and which kernel family runs it"""
    out, cin = [], 3
    for item in cfg:
        if item == "P":
            h, w = (h + 1) // 2, (w + 1) // 2
        else:
            out.append((2.0 * h * w * cin * 9 * item, "conv_wino" if cin >= 16 else "conv_direct"))
            cin = item
    return out


def synthetic_inputs(variant=0):
    """variant 0 = SURVEY §8d's inputs (image seed 555, ROI seed 556); variants 1.. = further images / proposal sets of the same
    distribution (the bench rotates four, so that consecutive steps do not upload and score the same bytes)"""
    rng = np.random.default_rng(555 + 1000 * variant)
    im = rng.random((3, H, W), dtype=np.float32)
    rng = np.random.default_rng(556 + 1000 * variant)
    boxes = np.zeros((0, 4), np.float32)
    while boxes.shape[0] < N_ROIS:  # centre uniform, log-uniform w,h in [16,600], clipped, area > 2
        c = rng.uniform([1, 1], [W, H], (2 * N_ROIS, 2))
        wh = np.exp(rng.uniform(np.log(16), np.log(600), (2 * N_ROIS, 2)))
        b = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)
        b = b[(b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 2]
        boxes = np.concatenate([boxes, b])[:N_ROIS]
    return im, boxes


def _np_params(P):
    return {k: ([t.numpy() for t in v] if isinstance(v, list) and v and hasattr(v[0], "numpy") else (v.numpy() if hasattr(v, "numpy") else v))
            for k, v in P.items()}


def cpu_path_torch(P, im, boxes, threads, n_rois):
    """The reference's per-image path on host cores with the strongest CPU kernels available here: PyTorch-CPU (oneDNN) for
    the trunk and the fc layers, the oracle's C ROI pool, numpy decode, and the reference's own compiled nms.c (single
    thread, as Tester_FRCNN.lua:117 runs it).  Returns seconds for (trunk, head on n_rois ROIs, nms on n_rois ROIs)."""
    import torch
    import torch.nn.functional as F
    from oracle import mpn_oracle as O
    from multipathnet_amd import models
    torch.set_num_threads(threads)
    Pn = _np_params(P)
    with torch.no_grad():
        t0 = time.time()
        x = torch.from_numpy(O.image_transform(im, **O.ROSS)).unsqueeze(0)
        li = 0
        for item in models.VGG16_CFG:
            if item == "P":
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            else:
                x = F.relu(F.conv2d(x, P["conv_w"][li], P["conv_b"][li], padding=1))
                li += 1
        feat = x[0].numpy()
        t_trunk = time.time() - t0
        t0 = time.time()
        b = boxes[:n_rois]
        pooled, _ = O.roi_pool(feat, O.project_im_rois(b, 1.0), 7, 7, 1.0 / 16)
        h = torch.from_numpy(pooled.reshape(b.shape[0], -1))
        h = F.relu(F.linear(h, P["fc6_w"], P["fc6_b"]))
        h = F.relu(F.linear(h, P["fc7_w"], P["fc7_b"]))
        logits = F.linear(h, P["cls_w"], P["cls_b"]).numpy()
        deltas = F.linear(h, P["bbox_w"], P["bbox_b"]).numpy()
        if Pn.get("bbox_mean") is not None:
            deltas = O.bbox_norm(deltas, Pn["bbox_mean"], Pn["bbox_std"])
        scores = O.softmax(logits)
        dec = O.clamp_boxes(O.bbox_decode(b, deltas), W, H)
        t_head = time.time() - t0
        t0 = time.time()
        nms = O.ref_nms if O.have_ref() else O.nms
        for j in range(1, scores.shape[1]):
            sb, _ = O.select_scored(scores, dec, j, -1.5)
            nms(sb, 0.3)
        t_nms = time.time() - t0
    return t_trunk, t_head, t_nms


def cpu_baseline(P, im, boxes, rois_sample):
    from oracle import mpn_oracle as O
    ncpu = os.cpu_count() or 1
    out = {"unit": "proposals/s", "kind": "port",
           "what": "PyTorch-CPU (oneDNN) conv2d / max_pool2d(ceil_mode) / linear + oracle C ROI pool + numpy decode + "
                   + ("the reference's own nms.c (compiled unmodified, 1 thread)" if O.have_ref() else "oracle NMS port")
                   + "; `value` is the BEST OF {16, 64} oneDNN threads (chosen after measuring both: `by_threads`), `cores` = the winner"}
    # several thread counts (oneDNN on one 600x1000 image does not scale to every core of a big host: 256 threads measured 9 s for
    # the trunk where 1 thread takes 2 s); the best one is the baseline, `cores` = the threads it used
    cpu_path_torch(P, im, boxes, min(ncpu, 16), 16)  # cold-start warm-up
    tried = {}
    for th_n in sorted({min(ncpu, c) for c in (16, 64)}):
        tt, th, tn = cpu_path_torch(P, im, boxes, th_n, N_ROIS)
        tried[th_n] = (tt + th + tn, tt, th, tn)
    best = min(tried, key=lambda k: tried[k][0])
    total, tt, th, tn = tried[best]
    out.update({"value": round(N_ROIS / total, 1), "cores": best, "host_cores": ncpu, "seconds_per_image": round(total, 3),
                "by_threads": {str(k): round(N_ROIS / v[0], 1) for k, v in tried.items()},
                "sample": "1 image 600x1000, all %d ROIs, after a warm-up: trunk %.2fs + ROI pool/fc/heads %.2fs + NMS %.2fs" % (N_ROIS, tt, th, tn)})
    # one thread: full trunk, head + NMS on a bounded ROI sample scaled to 1000 (rows / boxes are independent)
    tt1, th1, tn1 = cpu_path_torch(P, im, boxes, 1, rois_sample)
    sc = N_ROIS / float(rois_sample)
    total1 = tt1 + (th1 + tn1) * sc
    out["one_thread"] = {"value": round(N_ROIS / total1, 1), "cores": 1, "seconds_per_image": round(total1, 3),
                         "sample": "full trunk %.2fs + head on %d of %d ROIs scaled %.2fs + NMS scaled %.2fs" % (tt1, rois_sample, N_ROIS, th1 * sc, tn1 * sc)}
    # the plain C restatement (oracle/, OpenMP over all cores) — the checker itself, for reference
    Pn = _np_params(P)
    t0 = time.time()
    feat = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"])
    t_trunk = time.time() - t0
    t0 = time.time()
    b = boxes[:rois_sample]
    O.frcnn_head(feat, O.project_im_rois(b, 1.0), Pn, chunk=500)
    t_head = (time.time() - t0) * sc
    out["oracle_port"] = {"value": round(N_ROIS / (t_trunk + t_head + tn), 1), "cores": ncpu,
                          "sample": "oracle/mpn_oracle.c: full trunk %.2fs + head on %d ROIs scaled %.2fs (+ the NMS time above)" % (t_trunk, rois_sample, t_head)}
    return out


# A dataset-shaped stream (Tester:test loops over images of DIFFERENT sizes, Tester_FRCNN.lua:150-157; getImages rescales each to a 600-px
# short side with the long side capped at 1000, ImageDetect.lua:22-52): (H, W, proposals) -> the size the trunk sees
MIXED_SIZES = [(600, 1000, 1000),    # s = 1: the bench image's size
               (600, 800, 1000),     # s = 1, another aspect ratio
               (480, 640, 700),      # s = 1.25 -> 600 x 800 (bilinear up)
               (640, 480, 1000),     # portrait, s = 1.25 -> 800 x 600
               (400, 800, 300),      # s = 1.5 would give a 1200-px long side: capped, s = 1.25 -> 500 x 1000
               (1200, 1600, 1000)]   # s = 0.5 -> 600 x 800 (box-average down)


def mixed_size_inputs():
    """[(image [3,H,W] fp32 in [0,1), proposals [n,4] 1-based x1y1x2y2)] for MIXED_SIZES: seeded, same box distribution as synthetic_inputs"""
    out = []
    for i, (h, w, n) in enumerate(MIXED_SIZES):
        rng = np.random.default_rng(7000 + i)
        im = rng.random((3, h, w), dtype=np.float32)
        c = rng.uniform([1, 1], [w, h], (n, 2))
        wh = np.exp(rng.uniform(np.log(16), np.log(min(h, w)), (n, 2)))
        boxes = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [w, h, w, h]).astype(np.float32)
        out.append((im, boxes))
    return out


def more_boxes(boxes, n):
    """the synthetic proposal set cut / extended (same distribution) to n rows"""
    rng = np.random.default_rng(556)
    while boxes.shape[0] < n:
        boxes = np.concatenate([boxes, boxes[rng.permutation(boxes.shape[0])] * np.float32(0.97) + np.float32(1.0)])
    return np.clip(boxes[:n], 1, [W, H, W, H]).astype(np.float32)


def _oplist_flops(ops, h, w, wino_only=False):
    """algorithmic convolution FLOPs of an op list (include/mpn.h mpn_graph_op) on an h x w input; wino_only: only the layers an fp32
    graph TRUNK routes to the Winograd kernel (3x3 / stride 1 / pad 1 with >= 16 input channels: resnet.hip graph_run, fuse bit 5)"""
    dims, f = {0: (h, w)}, 0.0
    for o in ops:
        sh_, sw_ = dims[o["src"]]
        if o["kind"] == 3:
            oh, ow = sh_, sw_
        elif o["kind"] == 1 and o.get("ceil"):
            def cs(x, k, s, p):
                r = -(-(x + 2 * p - k) // s) + 1
                return r - 1 if (p > 0 and (r - 1) * s >= x + p) else r
            oh, ow = cs(sh_, o["kh"], o["sh"], o["ph"]), cs(sw_, o["kw"], o["sw"], o["pw"])
        else:
            oh, ow = (sh_ + 2 * o["ph"] - o["kh"]) // o["sh"] + 1, (sw_ + 2 * o["pw"] - o["kw"]) // o["sw"] + 1
        dims.setdefault(o["dst"], (oh, ow))
        if o["kind"] == 0 and (not wino_only or (o["kh"] == 3 and o["kw"] == 3 and o["sh"] == 1 and o["sw"] == 1 and o["ph"] == 1 and o["pw"] == 1 and o["cin"] >= 16)):
            f += 2.0 * o["cout"] * o["cin"] * o["kh"] * o["kw"] * oh * ow
    return f


def _groups(**kw):
    """kernel groups of a configuration keyed by the pipeline's profiling tag: (label, algorithmic FLOPs per image, of which on the
    Winograd kernel).  executed = algorithmic - winograd * (1 - 16/36): what the matrix pipe multiplies (padding waste not counted)."""
    return {tag: dict(label=label, alg=float(alg), wino=float(wino)) for tag, (label, alg, wino) in kw.items()}


def _cpu_baseline_towers(kind, P, im, boxes, n, C, K, sample, bf16=False):
    """cpu_baseline of the tower configurations (configs[2] / [3] / [4]) on this box's host cores: PyTorch-CPU (oneDNN; oracle/torch_ref.py —
    conv2d / max_pool2d / avg_pool2d / linear in fp32) for the trunk on the whole image and for the towers + integral heads on a BOUNDED
    ROI sample (rows are independent: scaled to all n ROIs), the oracle's C ROI pooling, numpy decode, and the reference's own nms.c
    (compiled unmodified, one thread as Tester_FRCNN.lua:117 runs it) on the sample's rows per class, scaled likewise."""
    import torch
    from oracle import mpn_oracle as O
    from oracle import torch_ref as T
    from multipathnet_amd import models
    th_n = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(th_n)
    b = np.ascontiguousarray(boxes[:sample])
    rois = O.project_im_rois(b, 1.0)
    fov = O.foveal(rois).reshape(-1, 4, 5)

    def run():
        t0 = time.time()
        if kind == "vgg":
            taps = {}
            T.vgg_trunk(O.image_transform(im, **O.ROSS), P, models.VGG16_CFG, taps)
            maps = [taps["conv5"], taps["conv4"], taps["conv3"]]
        elif kind == "resnet":
            feat = T.resnet_trunk(O.image_transform(im, **O.IMAGENET), P, bf16)
        else:
            feat = T.graph_trunk(O.image_transform(im, **O.INCEPTION), P, bf16)
        t_trunk = time.time() - t0
        t0 = time.time()
        fs = []
        if kind == "vgg":
            for Tw in P["towers"]:
                r = np.ascontiguousarray(fov[:, Tw["region"]])
                pools = [O.roi_pool(maps[m], r, 7, 7, (1.0 / 16) * (2 ** m))[0] if use else None for m, use in enumerate((1, Tw["use4"], Tw["use3"]))]
                fs.append(T.mpnet_tower(pools, Tw, True))
        elif kind == "resnet":
            for tw, rg in zip(P["head_towers"], P["head_regions"]):
                fs.append(T.resnet_tower(O.roi_pool(feat, np.ascontiguousarray(fov[:, rg]), 14, 14, 1.0 / 16)[0], tw, bf16))
        else:
            for tw, rg in zip(P["head_towers"], P["head_regions"]):
                fs.append(T.graph_tower(O.roi_pool(feat, np.ascontiguousarray(fov[:, rg]), 17, 17, 17.0 / 299.0)[0], tw, P, bf16))
        with torch.no_grad():
            logits = torch.nn.functional.linear(torch.cat(fs[:-1], 1), P["cls_w"], P["cls_b"]).numpy().reshape(sample, K, C)
            deltas = torch.nn.functional.linear(fs[-1], P["bbox_w"], P["bbox_b"]).numpy()
        if P.get("bbox_mean") is not None:
            deltas = O.bbox_norm(deltas, P["bbox_mean"], P["bbox_std"])
        scores = O.mean_over_k(np.stack([O.softmax(np.ascontiguousarray(logits[:, k])) for k in range(K)]))
        dec = O.clamp_boxes(O.bbox_decode(b, deltas), W, H)
        t_head = time.time() - t0
        t0 = time.time()
        # NMS is not linear in the row count, so it is timed at FULL size: all n proposal boxes per class (the overlap structure of the
        # real table), scored by cycling the sample's rows' scores
        nms = O.ref_nms if O.have_ref() else O.nms
        reps = -(-n // sample)
        for j in range(1, C):
            sb = np.concatenate([boxes[:n], np.tile(scores[:, j:j + 1], (reps, 1))[:n]], 1).astype(np.float32)
            nms(sb, 0.3)
        return t_trunk, t_head, time.time() - t0
    run()  # cold start
    tt, th, tn = run()
    sc = n / float(sample)
    total = tt + th * sc + tn
    return {"value": round(n / total, 1), "unit": "proposals/s", "cores": th_n, "host_cores": os.cpu_count() or 1, "kind": "port", "seconds_per_image": round(total, 3),
            "sample": "1 image 600x1000 after one warm-up pass: PyTorch-CPU fp32 trunk %.2fs + towers / heads on %d of %d ROIs scaled %.2fs + reference nms.c, %d classes x all %d "
                      "proposal boxes (scores cycled from the sample) %.2fs" % (tt, sample, n, th * sc, C - 1, n, tn)}


def _cfg_alexnet(models, args):  # BASELINE configs[0]
    n = 300
    G = models.synthetic_alexnet_params(n_classes=21, seed=557)
    net = models.AlexNetFRCNN(G, max_h=H, max_w=W, max_rois=n)
    flops = _oplist_flops(G["trunk_ops"], H, W) + n * (_oplist_flops(G["head_ops"], 6, 6) + 2.0 * 4096 * 105)
    groups = _groups(conv_direct=("trunk (conv1 as a GEMM over im2col rows, conv2 direct, conv3-5 Winograd, pools, LRN)", _oplist_flops(G["trunk_ops"], H, W),
                                  _oplist_flops(G["trunk_ops"], H, W, wino_only=True)),
                     fc6=("ROI pooling + fc6 + fc7 (row-invariant GEMMs)", n * _oplist_flops(G["head_ops"], 6, 6), 0.0),
                     heads=("cls + bbox GEMM", n * 2.0 * 4096 * 105, 0.0))

    def cpu(im, boxes):
        """the config's own 'CPU nn path': PyTorch-CPU conv2d(groups) / max_pool2d(ceil_mode) / local_response_norm / linear + oracle
        ROI pool + the reference's nms.c, all cores, one warm-up image"""
        import torch
        import torch.nn.functional as F
        from oracle import mpn_oracle as O
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        ops = G["trunk_ops"]

        def run():
            t0 = time.time()
            with torch.no_grad():
                t = torch.from_numpy(O.image_transform(im, **O.ROSS)).unsqueeze(0)
                i = 0
                while i < len(ops):
                    o = ops[i]
                    if o["kind"] == 0:
                        grp = [o]
                        while i + 1 < len(ops) and ops[i + 1]["kind"] == 0 and ops[i + 1]["dst"] == o["dst"]:
                            i += 1
                            grp.append(ops[i])
                        t = F.relu(F.conv2d(t, torch.cat([g["w"] for g in grp]), torch.cat([g["b"] for g in grp]), stride=o["sh"], padding=o["ph"], groups=len(grp)))
                    elif o["kind"] == 1:
                        t = F.max_pool2d(t, o["kh"], o["sh"], o["ph"], ceil_mode=bool(o.get("ceil")))
                    else:
                        t = F.local_response_norm(t, o["kh"], *o["lrn"])
                    i += 1
                pooled, _ = O.roi_pool(t[0].numpy(), O.project_im_rois(boxes, 1.0), 6, 6, 1.0 / 16)
                hh = torch.from_numpy(pooled.reshape(boxes.shape[0], -1))
                for o in G["head_ops"]:
                    hh = F.relu(F.linear(hh, o["w"].reshape(o["w"].shape[0], -1), o["b"]))
                logits = F.linear(hh, G["cls_w"], G["cls_b"]).numpy()
                deltas = O.bbox_norm(F.linear(hh, G["bbox_w"], G["bbox_b"]).numpy(), G["bbox_mean"], G["bbox_std"])
            sc, dec = O.softmax(logits), O.clamp_boxes(O.bbox_decode(boxes, deltas), W, H)
            nms = O.ref_nms if O.have_ref() else O.nms
            for j in range(1, sc.shape[1]):
                nms(O.select_scored(sc, dec, j, -1.5)[0], 0.3)
            return time.time() - t0
        run()
        dt = run()
        return {"value": round(n / dt, 1), "unit": "proposals/s", "cores": min(os.cpu_count() or 1, 64), "kind": "port", "seconds_per_image": round(dt, 3),
                "sample": "1 image 600x1000 x 300 ROIs after one warm-up image: PyTorch-CPU trunk / fc + oracle ROI pool + reference nms.c"}
    alg_bytes = (_oplist_bytes(G["trunk_ops"], H, W, 1, 4, G["trunk_tensor_c"]) + 4.0 * n * 256 * 36 + _oplist_bytes(G["head_ops"], 6, 6, n, 4, G["head_tensor_c"])
                 + 4.0 * (4096 * 105 + n * 105))
    return dict(params=G, net=net, n_rois=n, flops=flops, dtype="f32", cpu_baseline=cpu, groups=groups, key="c1", alg_bytes=alg_bytes,
                metric="proposals/sec (300 ROIs, 600x1000 img) AlexNet Fast R-CNN [BASELINE configs[0]; not the headline metric]",
                workload="AlexNet / CaffeNet Fast R-CNN (models/alexnet.lua), 1 image 600x1000 x 300 ROIs per GPU per step, 21 classes")


def _cfg_vgg_frcnn_n(models, args):  # the HEADLINE model at another proposal count: `--config c2 --rois 2000` (auxiliary line, never the headline)
    n = int(args.rois)
    split3 = getattr(args, "fc_arith", "fp32") == "split3"
    P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=N_CLASSES, seed=557)
    net = models.FastRCNN(P, max_h=H, max_w=W, max_rois=n, fc_arith=1 if split3 else 0)
    cf = conv_flops(models.VGG16_CFG, H, W)
    wino = sum(f for f, v in cf if v == "conv_wino")
    head = 2.0 * (25088 * 4096 + 4096 * 4096 + 4096 * 5 * N_CLASSES)
    groups = _groups(conv_wino=("trunk, 12 Winograd F(2x2,3x3) layers", wino, wino), conv_direct=("conv1_1", sum(f for f, v in cf if v == "conv_direct"), 0.0),
                     fc6=("fc6", n * 2.0 * 25088 * 4096, 0.0), fc7=("fc7", n * 2.0 * 4096 * 4096, 0.0), heads=("cls + bbox GEMM", n * 2.0 * 4096 * 5 * N_CLASSES, 0.0))
    # compulsory HBM bytes of one image, every layer reading its input and writing its output once (fp32), weights once
    trunk_b = _vgg_trunk_bytes(models.VGG16_CFG, H, W)
    head_b = 4.0 * (n * 25088 * 2 + 25088 * 4096 + n * 4096 * 2 + 4096 * 4096 + n * 4096 * 2 + 4096 * 5 * N_CLASSES + n * 5 * N_CLASSES)
    if split3:  # VERDICT r5 task 2: an AUXILIARY line with its own metric string and dtype; the headline stays pure fp32 MFMA
        return dict(params=P, net=net, n_rois=n, flops=sum(f for f, _ in cf) + n * head, dtype="f32 (fc6: 3-plane bf16 split, 6 products, fp32 accumulate)", groups=groups,
                    key="c2_split3_n%d" % n, cpu_baseline=None, alg_bytes=trunk_b + head_b,
                    metric="proposals/sec (%d ROIs, 600x1000 img) VGG-16 Fast R-CNN, fc6 on the bf16 matrix pipe as an exact three-plane split with fp32 accumulation "
                           "[MPN_FC_SPLIT3; auxiliary line, not the headline metric: the headline is pure fp32 MFMA]" % n,
                    workload="VGG-16 Fast R-CNN, 1 image 600x1000 x %d ROIs per GPU per step, 21 classes, NMS 0.3, top-100; fc6 = v_mfma_f32_32x32x16_bf16 x 6 plane products" % n)
    return dict(params=P, net=net, n_rois=n, flops=sum(f for f, _ in cf) + n * head, dtype="f32", groups=groups, key="c2_n%d" % n, cpu_baseline=None,
                alg_bytes=trunk_b + head_b,
                metric="proposals/sec (%d ROIs, 600x1000 img) VGG-16 Fast R-CNN [the headline model at the proposal count of scripts/eval_fastrcnn_voc2007.sh; "
                       "auxiliary line, not the headline metric]" % n,
                workload="VGG-16 Fast R-CNN, 1 image 600x1000 x %d ROIs per GPU per step, 21 classes, NMS 0.3, top-100" % n)


def _vgg_trunk_bytes(cfg, h, w):
    """fp32 bytes a VGG trunk must move when every layer reads its input map and writes its output map once (a fused ceil-mode pool writes
    the pooled map only) and reads its weights once"""
    b, cin, li = 0.0, 3, 0
    items = list(cfg)
    for i, item in enumerate(items):
        if item == "P":
            continue
        pooled = i + 1 < len(items) and items[i + 1] == "P"
        oh, ow = ((h + 1) // 2, (w + 1) // 2) if pooled else (h, w)
        b += 4.0 * (cin * h * w + item * oh * ow + item * cin * 9 + item)
        cin, h, w = item, oh, ow
    return b


def _oplist_bytes(ops, h, w, batch, esz, tensor_c=None):
    """HBM bytes of an op list (include/mpn.h mpn_graph_op) on `batch` maps of h x w.  tensor_c = None: the LAYER-WISE count — every op reads
    its input channels and writes its output channels once (element size esz), convolutions read their weights once per launch.  tensor_c =
    the list's tensor channel counts: the TENSOR-LEVEL floor — every tensor of the list is written once and read once (however many ops
    consume it: an Inception block's four branches share one read of their input), weights once."""
    if tensor_c is not None:
        dims, wb = {0: (h, w)}, 0.0
        for o in ops:
            sh_, sw_ = dims[o["src"]]
            if o["kind"] == 3:
                oh, ow = sh_, sw_
            elif o["kind"] == 1 and o.get("ceil"):
                def cs2(x, k, s_, p_):
                    r = -(-(x + 2 * p_ - k) // s_) + 1
                    return r - 1 if (p_ > 0 and (r - 1) * s_ >= x + p_) else r
                oh, ow = cs2(sh_, o["kh"], o["sh"], o["ph"]), cs2(sw_, o["kw"], o["sw"], o["pw"])
            else:
                oh, ow = (sh_ + 2 * o["ph"] - o["kh"]) // o["sh"] + 1, (sw_ + 2 * o["pw"] - o["kw"]) // o["sw"] + 1
            dims.setdefault(o["dst"], (oh, ow))
            if o["kind"] == 0:
                wb += esz * o["cout"] * o["cin"] * o["kh"] * o["kw"]
        return wb + sum(batch * esz * tensor_c[t] * hh * ww * (1.0 if t == 0 else 2.0) for t, (hh, ww) in dims.items())
    dims, b = {0: (h, w)}, 0.0
    for o in ops:
        sh_, sw_ = dims[o["src"]]
        if o["kind"] == 3:
            oh, ow = sh_, sw_
        elif o["kind"] == 1 and o.get("ceil"):
            def cs(x, k, s, p):
                r = -(-(x + 2 * p - k) // s) + 1
                return r - 1 if (p > 0 and (r - 1) * s >= x + p) else r
            oh, ow = cs(sh_, o["kh"], o["sh"], o["ph"]), cs(sw_, o["kw"], o["sw"], o["pw"])
        else:
            oh, ow = (sh_ + 2 * o["ph"] - o["kh"]) // o["sh"] + 1, (sw_ + 2 * o["pw"] - o["kw"]) // o["sw"] + 1
        dims.setdefault(o["dst"], (oh, ow))
        cout = o["cout"] if o["kind"] == 0 else o["cin"]
        b += batch * esz * (o["cin"] * sh_ * sw_ + cout * oh * ow)
        if o["kind"] == 0:
            b += esz * o["cout"] * o["cin"] * o["kh"] * o["kw"]
    return b


def _cfg_vgg_mpn(models, args):  # BASELINE configs[2]
    n = 1000
    P = models.synthetic_mpnet_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=81, n_integral=6, seed=557)
    split3 = getattr(args, "fc_arith", "fp32") == "split3"
    net = models.MultiPathNet(P, max_h=H, max_w=W, max_rois=n, fc_arith=1 if split3 else 0)
    flops = (367.74e9 + 5 * n * 2.0 * (25088 * 4096 + 4096 * 4096) + 49 * n * 2.0 * 512 * (1280 + 1024 + 1024 + 512 + 1280)
             + n * 2.0 * (16384 * 486 + 4096 * 324))
    cf = conv_flops(models.VGG16_CFG, H, W)
    wino = sum(f for f, v in cf if v == "conv_wino")
    groups = _groups(conv_wino=("trunk, 12 Winograd F(2x2,3x3) layers", wino, wino), conv_direct=("conv1_1", sum(f for f, v in cf if v == "conv_direct"), 0.0),
                     fc6=("fc6 x 5 towers", 5 * n * 2.0 * 25088 * 4096, 0.0), fc7=("fc7 x 5 towers", 5 * n * 2.0 * 4096 * 4096, 0.0),
                     heads=("1x1 mix GEMMs x 5 + integral classifiers + box regressor", 49 * n * 2.0 * 512 * (1280 + 1024 + 1024 + 512 + 1280) + n * 2.0 * (16384 * 486 + 4096 * 324), 0.0))

    def cpu(im, boxes):
        return _cpu_baseline_towers("vgg", P, im, boxes, n, 81, 6, sample=32)
    tow_c = (1280, 1024, 1024, 512, 1280)
    alg_bytes = (_vgg_trunk_bytes(models.VGG16_CFG, H, W)
                 + sum(4.0 * (n * 49 * c * 2 + c * 512 + n * 25088 * 2 + 25088 * 4096 + n * 4096 * 2 + 4096 * 4096 + n * 4096) for c in tow_c)
                 + 4.0 * (16384 * 486 + 4096 * 324 + n * 20480 + n * (486 + 324)))
    if split3:
        return dict(params=P, net=net, n_rois=n, flops=flops, dtype="f32 (fc6 / fc7: 3-plane bf16 split, 6 products, fp32 accumulate)", groups=groups, key="c3_split3",
                    cpu_baseline=None, alg_bytes=alg_bytes,
                    metric="proposals/sec (1000 ROIs, 600x1000 img) VGG-16 MultiPathNet, the towers' fc6 / fc7 on the bf16 matrix pipe as exact three-plane splits "
                           "with fp32 accumulation [MPN_FC_SPLIT3; auxiliary line: BASELINE configs[2] in plain fp32 is the --config c3 line]",
                    workload="VGG-16 MultiPathNet (4 foveal towers + box tower, conv3/4/5 skip pooling, K = 6 integral classifiers, 81 classes), 1000 ROIs; fc6 / fc7 = 6 bf16 plane products")
    return dict(params=P, net=net, n_rois=n, flops=flops, dtype="f32", groups=groups, key="c3", cpu_baseline=cpu, alg_bytes=alg_bytes,
                metric="proposals/sec (1000 ROIs, 600x1000 img) VGG-16 MultiPathNet [BASELINE configs[2]; not the headline metric]",
                workload="VGG-16 MultiPathNet (4 foveal towers + box tower, conv3/4/5 skip pooling, K = 6 integral classifiers, 81 classes), 1000 ROIs")


def _cfg_resnet_mpn(models, args):  # BASELINE configs[3]
    n = 1000
    bf16 = args.dtype == "bf16"
    R = models.synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=557)
    net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=n, bf16=bf16)

    def blocks_flops(blocks, h, w, wino_only=False):
        f = 0.0
        for b in blocks:
            bh, bw = h, w
            if b["shortcut"] is not None and not wino_only:
                ws, _, st = b["shortcut"]
                f += 2.0 * ws.shape[0] * ws.shape[1] * ((h - 1) // st + 1) * ((w - 1) // st + 1)
            for (wt, _, st, pd) in b["convs"]:
                k = wt.shape[2]
                bh, bw = (bh + 2 * pd - k) // st + 1, (bw + 2 * pd - k) // st + 1
                if not wino_only or (k == 3 and st == 1 and pd == 1 and wt.shape[1] >= 16):  # resnet.hip fuse bit 7: the head's 3x3 / 1 / 1 layers on the mosaic
                    f += 2.0 * wt.shape[0] * wt.shape[1] * k * k * bh * bw
            h, w = bh, bw
        return f
    def blocks_bytes(blocks, h, w, batch, esz, floor=False):
        """floor = False, LAYER-WISE: every convolution reads its input and writes its output once (+ weights once per launch), the residual operand
        is read once.  floor = True, TENSOR-LEVEL: every tensor a block produces is written once and read once (the block input's two consumers —
        first convolution and shortcut — share one read), weights once."""
        b = 0.0
        if floor:
            for blk in blocks:
                bh, bw = h, w
                if blk["shortcut"] is not None:
                    ws, _, st = blk["shortcut"]
                    b += batch * esz * 2.0 * ws.shape[0] * ((h - 1) // st + 1) * ((w - 1) // st + 1) + esz * ws.numel()
                for (wt, _, st, pd) in blk["convs"]:
                    k = wt.shape[2]
                    bh, bw = (bh + 2 * pd - k) // st + 1, (bw + 2 * pd - k) // st + 1
                    b += batch * esz * 2.0 * wt.shape[0] * bh * bw + esz * wt.numel()
                h, w = bh, bw
            return b
        for blk in blocks:
            bh, bw = h, w
            cin0 = blk["convs"][0][0].shape[1]
            if blk["shortcut"] is not None:
                ws, _, st = blk["shortcut"]
                b += batch * esz * (ws.shape[1] * h * w + ws.shape[0] * ((h - 1) // st + 1) * ((w - 1) // st + 1)) + esz * ws.numel()
            cout_last = blk["convs"][-1][0].shape[0]
            for (wt, _, st, pd) in blk["convs"]:
                k = wt.shape[2]
                oh, ow = (bh + 2 * pd - k) // st + 1, (bw + 2 * pd - k) // st + 1
                b += batch * esz * (wt.shape[1] * bh * bw + wt.shape[0] * oh * ow) + esz * wt.numel()
                bh, bw = oh, ow
            b += batch * esz * cout_last * bh * bw   # the residual read by the block's last convolution
            h, w = bh, bw
        return b
    h1, w1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    h2, w2 = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1
    n_tow = len(R["head_towers"])
    esz = 2 if bf16 else 4
    c_feat = R["head_blocks"][0]["convs"][0][0].shape[1]
    tail4 = 4.0 * ((n_tow - 1) * 2048 * 6 * 81 + 2048 * 4 * 81 + n * n_tow * 2048)
    stem4 = 4.0 * 3 * H * W + esz * (64 * h1 * w1 * 2 + 64 * h2 * w2 * 2)
    alg_bytes = (stem4 + blocks_bytes(R["trunk_blocks"], h2, w2, 1, esz, True)
                 + n_tow * (esz * 2.0 * n * c_feat * 196 + blocks_bytes(R["head_blocks"], 14, 14, n, esz, True) + 4.0 * n * 2048) + tail4)
    layerwise_bytes = (stem4 + blocks_bytes(R["trunk_blocks"], h2, w2, 1, esz)
                       + n_tow * (esz * n * c_feat * 196 + blocks_bytes(R["head_blocks"], 14, 14, n, esz) + 4.0 * n * 2048) + tail4)
    f_trunk = 2.0 * 64 * 3 * 49 * h1 * w1 + blocks_flops(R["trunk_blocks"], h2, w2)
    f_tow = n * blocks_flops(R["head_blocks"], 14, 14) * n_tow
    f_heads = n * 2.0 * ((n_tow - 1) * 2048 * 6 * 81 + 2048 * 4 * 81)
    flops = f_trunk + f_tow
    groups = _groups(conv_direct=("trunk: conv1, max-pool, layer1-3 on the image", f_trunk, 0.0),
                     fc6=("per-ROI towers: ROI pooling, 5 x layer4, average pool", f_tow, 0.0 if bf16 else n * blocks_flops(R["head_blocks"], 14, 14, True) * n_tow),
                     heads=("integral classifiers + box regressor", f_heads, 0.0))

    def cpu(im, boxes):
        return _cpu_baseline_towers("resnet", R, im, boxes, n, 81, 6, sample=16, bf16=False)
    return dict(params=R, net=net, n_rois=n, flops=flops, dtype="bf16" if bf16 else "f32", groups=groups, key="c4_bf16" if bf16 else "c4", cpu_baseline=cpu, alg_bytes=alg_bytes, layerwise_bytes=layerwise_bytes,
                metric="proposals/sec (1000 ROIs, 600x1000 img) ResNet-50 MultiPathNet [BASELINE configs[3]; not the headline metric]",
                workload="ResNet-50 with MultiPathNet towers (this library's extension of models/resnet.lua: 5 layer4 towers over Foveal regions, K = 6, "
                         "81 classes), 1000 ROIs, %s" % ("bf16 activations / weights, fp32 accumulate" if bf16 else "fp32"))


def _cfg_inception_mpn(models, args):  # BASELINE configs[4]
    n = 2000
    G = models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557)
    net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=n, bf16=True)
    flops = _oplist_flops(G["trunk_ops"], H, W) + n * _oplist_flops(G["head_ops"], 17, 17) * len(G["head_towers"])
    n_tow = len(G["head_towers"])
    groups = _groups(conv_direct=("trunk: stem + Mixed_5b..6e on the image", _oplist_flops(G["trunk_ops"], H, W), 0.0),
                     fc6=("per-ROI towers: ROI pooling, 5 x Mixed_7a..7c, average pool", n * _oplist_flops(G["head_ops"], 17, 17) * n_tow, 0.0),
                     heads=("integral classifiers + box regressor", n * 2.0 * ((n_tow - 1) * 2048 * 6 * 81 + 2048 * 4 * 81), 0.0))

    def cpu(im, boxes):
        return _cpu_baseline_towers("graph", G, im, boxes, n, 81, 6, sample=16)
    c_feat5 = G["trunk_tensor_c"][G["feat_tensor"]]
    c_out5 = G["bbox_w"].shape[1]
    tail5 = 4.0 * ((n_tow - 1) * c_out5 * 6 * 81 + c_out5 * 4 * 81 + n * n_tow * c_out5)
    # tensor-level floor: every tensor written once and read once (the ROI-pooled tensor = tensor 0 of the head list: + its one write), weights once
    alg_bytes = (4.0 * 3 * H * W + _oplist_bytes(G["trunk_ops"], H, W, 1, 2, G["trunk_tensor_c"])
                 + n_tow * (2.0 * n * c_feat5 * 289 + _oplist_bytes(G["head_ops"], 17, 17, n, 2, G["head_tensor_c"]) + 4.0 * n * c_out5) + tail5)
    layerwise_bytes = (4.0 * 3 * H * W + _oplist_bytes(G["trunk_ops"], H, W, 1, 2)
                       + n_tow * (2.0 * n * c_feat5 * 289 + _oplist_bytes(G["head_ops"], 17, 17, n, 2) + 4.0 * n * c_out5) + tail5)
    return dict(params=G, net=net, n_rois=n, flops=flops, dtype="bf16", groups=groups, key="c5", cpu_baseline=cpu, alg_bytes=alg_bytes, layerwise_bytes=layerwise_bytes,
                metric="proposals/sec (2000 ROIs, 600x1000 img) Inception-v3 MultiPathNet bf16 [BASELINE configs[4]; not the headline metric]",
                workload="Inception-v3 with MultiPathNet towers (this library's extension of models/inceptionv3.lua: 5 Mixed_7a..7c towers over Foveal "
                         "regions, K = 6, 81 classes), 2000 ROIs, bf16 activations / weights, fp32 accumulate")


OTHER_CONFIGS = {"c1": _cfg_alexnet, "c3": _cfg_vgg_mpn, "c4": _cfg_resnet_mpn, "c5": _cfg_inception_mpn}


def _gpu_bdf(dev_index):
    import torch
    pr = torch.cuda.get_device_properties(dev_index)
    return "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)


class ClockSampler(object):
    """GPU clock / power beside a timed leg.  Source 1: sysfs hwmon of the device's PCI function (freq1_input = sclk in Hz,
    power1_average / power1_input in uW), read every 50 ms by a thread.  Source 2 (no readable sysfs): `rocm-smi --showclocks --showpower
    --json` polled by the same thread (each poll is a subprocess, so a few samples per second).  Never raises; reports what it could read."""

    def __init__(self, dev_index):
        import glob
        import threading
        self.samples, self.source, self._stop, self._thr = [], None, threading.Event(), None
        self._f_clk = self._f_pow = None
        try:
            base = "/sys/bus/pci/devices/" + _gpu_bdf(dev_index)
            for hw in sorted(glob.glob(base + "/hwmon/hwmon*")):
                if os.path.exists(hw + "/freq1_input"):
                    self._f_clk = hw + "/freq1_input"
                for name in ("power1_average", "power1_input"):
                    if self._f_pow is None and os.path.exists(hw + "/" + name):
                        self._f_pow = hw + "/" + name
            if self._f_clk or self._f_pow:
                self.source = "sysfs hwmon (%s%s)" % (os.path.basename(self._f_clk or ""), (", " + os.path.basename(self._f_pow)) if self._f_pow else "")
        except Exception:  # noqa: BLE001
            pass
        self._dev = dev_index
        self._smi = None
        if self.source is None:
            for exe in ("/opt/rocm/bin/rocm-smi", "rocm-smi"):
                try:
                    subprocess.check_output([exe, "--showclocks", "--showpower", "--json"], stderr=subprocess.DEVNULL, timeout=10)
                    self._smi, self.source = exe, "rocm-smi --showclocks --showpower --json (subprocess polls)"
                    break
                except Exception:  # noqa: BLE001
                    continue
        self._threading = threading

    def _read_sysfs(self):
        clk = pw = None
        try:
            if self._f_clk:
                clk = float(open(self._f_clk).read().strip()) / 1e6
            if self._f_pow:
                pw = float(open(self._f_pow).read().strip()) / 1e6
        except Exception:  # noqa: BLE001
            pass
        return clk, pw

    def _read_smi(self):
        import re
        clk = pw = None
        try:
            j = json.loads(subprocess.check_output([self._smi, "--showclocks", "--showpower", "--json"], stderr=subprocess.DEVNULL, timeout=10))
            card = j.get("card%d" % self._dev) or next(iter(j.values()))
            for k, v in card.items():
                kl = k.lower()
                m = re.search(r"([0-9.]+)", str(v))
                if not m:
                    continue
                if "sclk" in kl and "clock" in kl and clk is None:
                    clk = float(m.group(1))
                if "power" in kl and ("average" in kl or "socket" in kl or "current" in kl) and pw is None:
                    pw = float(m.group(1))
        except Exception:  # noqa: BLE001
            pass
        return clk, pw

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(),) + (self._read_sysfs() if self._smi is None else self._read_smi()))
            self._stop.wait(0.05 if self._smi is None else 0.3)

    def start(self):
        if self.source is not None:
            self._thr = self._threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=15)

    def summary(self):
        def st(vals):
            vals = [v for v in vals if v is not None]
            return None if not vals else {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        return {"source": self.source or "none readable (no sysfs hwmon for the device, no rocm-smi)", "samples": len(self.samples),
                "sclk_mhz": st([s_[1] for s_ in self.samples]), "power_w": st([s_[2] for s_ in self.samples])}


def pin_to_gpu_numa_node(dev_index):
    """CPU affinity of this rank = the CPUs local to its GPU (sysfs local_cpulist of the device's PCI function).  Best effort:
    returns a short description for the JSON line, never raises."""
    try:
        bdf = _gpu_bdf(dev_index)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read().strip())
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if node < 0 or not cpus:
            return "none (no NUMA node reported for %s)" % bdf
        os.sched_setaffinity(0, cpus)
        return "NUMA node %d of %s (%d cpus)" % (node, bdf, len(cpus))
    except Exception as e:  # noqa: BLE001
        return "none (%s)" % str(e).splitlines()[-1][:80]


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: run N ranks of this script under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def latency_mode(args, torch, dist, models, parallel, comm, rank, world, dev, share_gpu, affinity):
    """--mode latency: ONE image's proposals and classes sharded over the ranks (mpn_frcnn_test_one_sharded; replaces
    ModelParallelTable.lua:195-242 for a single image).  A step = host image -> H2D -> trunk (every rank) -> ROI head on this rank's
    1/N of the proposals -> all-gather of decoded rows -> NMS of this rank's 1/N of the classes -> all-gather of kept tables ->
    top-100, with a device synchronisation after every step (latency, not throughput: nothing of image i+1 overlaps image i)."""
    if comm is None and not share_gpu:
        raise SystemExit("bench.py --mode latency: the C-ABI RCCL communicator did not come up")
    im_np, boxes_np = synthetic_inputs()
    if args.config == "c3":  # VGG-16 MultiPathNet: 5 towers — the per-ROI head is 88 % of the image, which is what this mode shards
        other = OTHER_CONFIGS["c3"](models, args)
        net, n_rois, n_classes, model_name = other["net"], other["n_rois"], 81, "VGG-16 MultiPathNet (5 towers, K = 6, 81 classes)"
        boxes_np = more_boxes(boxes_np, n_rois)
    elif args.config == "c2":
        P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=N_CLASSES, seed=557)
        net, n_rois, n_classes, model_name = models.FastRCNN(P, max_h=H, max_w=W, max_rois=N_ROIS), N_ROIS, N_CLASSES, "VGG-16 Fast R-CNN"
    else:
        raise SystemExit("bench.py --mode latency supports --config c2 (default) and c3")
    im_host, boxes_host = torch.from_numpy(im_np).pin_memory(), torch.from_numpy(boxes_np).pin_memory()
    im_dev, boxes_dev = torch.empty(im_host.shape, device=dev), torch.empty(boxes_host.shape, device=dev)

    def upload():
        im_dev.copy_(im_host, non_blocking=True)
        boxes_dev.copy_(boxes_host, non_blocking=True)

    def timed(fn, with_upload=True):
        for _ in range(args.warmup):
            upload(); fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if with_upload:
                upload()
            fn()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt / args.steps * 1e3

    if share_gpu:
        # TEST MODE (MPN_BENCH_SHARE_GPU=1): every rank on device 0 — RCCL refuses two ranks on one device, so the two all-gathers of
        # mpn_frcnn_test_one_sharded go through the gloo group, records staged on the host; the three device phases are the C ABI's own
        # (mpn_frcnn_shard_head / _nms / _finish).  Exercises the launcher, the sharding arithmetic and the reductions at this world size.
        rr, cr = net.shard_record_floats(n_rois, world)

        def gather_rows(mine, n_floats):
            parts = [torch.empty(n_floats, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, mine.detach().cpu().contiguous())
            return torch.stack(parts).to(dev)

        def sharded():
            rows = net.shard_head(im_dev, boxes_dev, rank, world)
            rows_all = gather_rows(rows, rr)
            crec = net.shard_nms(rows_all, n_rois, rank, world)
            class_all = gather_rows(crec, cr)
            return net.shard_finish(class_all, n_rois, world)

        def gather_final(d, n):
            rec = parallel.pack_record(d, n, d.size(0)).cpu()
            return parallel.gather_detections(rec).to(dev)
    else:
        def sharded():
            return net.test_one_sharded(comm, im_dev, boxes_dev)

        def gather_final(d, n):
            return comm.gather_dets(d, n)
    ms = timed(sharded)
    ms_res = timed(sharded, with_upload=False)
    # outside the timed loops: every rank must hold the SAME final detections (one extra gather of the ~11-KB record), checked on rank 0
    dets_f, n_f = sharded()
    allrec = gather_final(dets_f, n_f)
    torch.cuda.synchronize()
    identical = bool(all(torch.equal(allrec[r], allrec[0]) for r in range(world))) and int(allrec[0, -1].item()) > 0
    if rank == 0 and not identical:
        raise SystemExit("bench.py --mode latency: the ranks ended with different detections")
    out = {"metric": "per-image latency (%d ROIs, 600x1000 img) %s, proposals + classes of ONE image sharded over the GPUs "
                     "[latency mode; not the headline metric]" % (n_rois, model_name),
           "value": round(ms, 4), "unit": "ms/image", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s, ONE image 600x1000 x %d ROIs per step for the whole job, %d classes, NMS 0.3, top-100; host image + "
                                  "boxes uploaded inside the step; device synchronisation after every step" % (model_name, n_rois, n_classes),
                      "parallelism": ("every rank: trunk on the whole image; ROI head on 1/%d of the proposals; RCCL all-gather of decoded rows; NMS of 1/%d of "
                                      "the classes; RCCL all-gather of kept tables (mpn_frcnn_test_one_sharded through the C ABI); cpu affinity: %s"
                                      % (world, world, affinity))
                                     + ("; TEST MODE MPN_BENCH_SHARE_GPU=1: all ranks share device 0 and both all-gathers go through gloo (host-staged) — NOT a "
                                        "multi-GPU measurement" if share_gpu else "")},
           "proposals_per_s": round(n_rois / (ms * 1e-3), 1), "ms_inputs_resident": round(ms_res, 4),
           "ranks": {"rccl_ranks": comm.rccl_ranks if comm is not None else 0, "final_detections_identical_on_all_ranks": identical,
                     "n_detections": int(allrec[0, -1].item())}}
    if world == 1:
        out["unsharded_ms"] = round(timed(lambda: net.test_one_async(im_dev, boxes_dev)), 4)  # mpn_frcnn_test_one under the same protocol
        G = max(2, args.emulate_world)
        rr, cr = net.shard_record_floats(n_rois, G)
        rows_all = torch.empty((G, rr), dtype=torch.float32, device=dev)
        class_all = torch.empty((G, cr), dtype=torch.float32, device=dev)
        for r in range(G):
            net.shard_head(im_dev, boxes_dev, r, G, out=rows_all[r])
        for r in range(G):
            net.shard_nms(rows_all, n_rois, r, G, out=class_all[r])

        def rank0_share():
            net.shard_head(im_dev, boxes_dev, 0, G, out=rows_all[0])
            net.shard_nms(rows_all, n_rois, 0, G, out=class_all[0])
            net.shard_finish(class_all, n_rois, G)
        proj_ms = timed(rank0_share)
        net.set_profiling(True)
        net.get_profile(reset=True)
        for _ in range(args.steps):
            rank0_share()
        torch.cuda.synchronize()
        prof = net.get_profile(reset=True)
        net.set_profiling(False)
        out["projected"] = {"world": G, "rank0_compute_ms": round(proj_ms, 4),
                            "kernel_groups_ms": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]},
                            "what": "ONE GPU running rank 0's share of a %d-rank world (trunk + %d of %d ROIs + %d of %d classes + top-100) with the "
                                    "other ranks' records precomputed: the two all-gathers (%.0f KB + %.0f KB in total) are NOT included — a "
                                    "projection, not a multi-GPU measurement" % (G, -(-n_rois // G), n_rois, -(-(n_classes - 1) // G), n_classes - 1,
                                                                                   rr * G * 4 / 1024.0, cr * G * 4 / 1024.0)}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def power_sensitivity(torch, models, P, im_dev, boxes_dev, dev_index, prof):
    """VERDICT r4 weak #8 / task 6a: fc6 (M = 1000, K = 25088, N = 4096; 205.5 GFLOP) ALONE, issued back to back for ~1 s, (a) on the operand the
    pipeline hands it (ROI-pooled post-ReLU conv5 features, the model's fc6 weights) and (b) on dense random operands, clock and power sampled
    beside each — next to the same GEMM's time INSIDE the pipeline (HIP events of the profiled leg), where it follows the trunk's phases.  The
    back-to-back legs run in the DEBUG flavour of the library (same sources and kernels; mpn_debug_bench_fc6 / mpn_debug_bench_linear)."""
    import ctypes as C
    from multipathnet_amd import _lib
    out = {"what": "fc6 alone, back to back, ~1 s per leg (debug-flavour library, same kernel: gemm_c8_pf_kernel); in_pipeline = the profiled leg's fc6 group",
           "gflop": round(2.0 * N_ROIS * 25088 * 4096 / 1e9, 2)}
    try:
        fl = 2.0 * N_ROIS * 25088 * 4096
        if prof.get("fc6", (0, 0))[1]:
            ms_in = prof["fc6"][0] / prof["fc6"][1]
            out["in_pipeline"] = {"us": round(ms_in * 1e3, 1), "frac_of_fp32_mfma_peak": round(fl / (ms_in * 1e-3) / FP32_MFMA_PEAK, 4)}
        with _lib.debug_hooks() as dlib:
            netd = models.FastRCNN(P, max_h=H, max_w=W, max_rois=N_ROIS)
            netd.detect(im_dev, boxes_dev)
            torch.cuda.synchronize()
            ms = C.c_float()
            legs = (("pipeline_operand", lambda it: dlib.mpn_debug_bench_fc6(netd._h, it, C.byref(ms))),
                    ("dense_random_operand", lambda it: dlib.mpn_debug_bench_linear(N_ROIS, 25088, 4096, it, C.byref(ms))))
            for name, fn in legs:
                _lib.check(fn(20), name)
                iters = max(50, int(1.0 / max(ms.value * 1e-3, 1e-4)))
                sampler = ClockSampler(dev_index).start()
                _lib.check(fn(iters), name)
                sampler.stop()
                leg = {"us": round(ms.value * 1e3, 1), "iters": iters, "frac_of_fp32_mfma_peak": round(fl / (ms.value * 1e-3) / FP32_MFMA_PEAK, 4)}
                leg.update(sampler.summary())
                out[name] = leg
            del netd
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001  (an auxiliary leg never takes the headline line down)
        out["error"] = str(e).splitlines()[-1][:200]
    return out


def mixed_sizes_leg(torch, dist, models, P, dev, dev_index, world, rank, seconds, pin=None):
    """VERDICT r4 task 6d: a sustained leg over a dataset-shaped stream — bench.MIXED_SIZES, six (image size, proposal count) pairs in rotation,
    each image rescaled ON THE DEVICE by getImages' rule (600-px short side, 1000-px cap: s = 1, 1.25, 0.5 ...), host-fed through
    mpn_frcnn_test_one_pipelined_host like the headline step.  Every size change re-lays the activation halos.  Its own key, never the metric."""
    stream = mixed_size_inputs()
    net = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=N_ROIS, scale=600, max_size=1000)
    if pin is None:
        pin = [(torch.from_numpy(i).pin_memory(), torch.from_numpy(b).pin_memory()) for i, b in stream]
    n_rois = [b.shape[0] for _, b in stream]

    def run(n_steps, start=0):
        for t in range(n_steps):
            i, b = pin[(start + t + rank) % len(pin)]
            net.test_one_pipelined_host(i, b)
        net.flush()
        torch.cuda.synchronize()
    run(2 * len(pin))
    t0 = time.perf_counter()
    run(len(pin))
    per_round = time.perf_counter() - t0
    rounds = max(3, int(np.ceil(seconds / per_round)))
    if world > 1:
        dist.barrier()
    sampler = ClockSampler(dev_index).start()
    t0 = time.perf_counter()
    run(rounds * len(pin))
    dt = time.perf_counter() - t0
    sampler.stop()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"what": "host-fed pipelined steps over MIXED_SIZES in rotation (a size change at EVERY step), getImages' rescale on the device; MAX over ranks",
           "sizes": ["%dx%d x %d ROIs" % (h, w, n) for h, w, n in MIXED_SIZES], "images": rounds * len(pin), "seconds": round(dt, 3),
           "images_per_s": round(world * rounds * len(pin) / dt, 2), "value": round(world * rounds * sum(n_rois) / dt, 1), "unit": "proposals/s",
           "ms_per_image": round(dt / (rounds * len(pin)) * 1e3, 4)}
    out.update(sampler.summary())
    del net
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rois", type=int, default=250)
    ap.add_argument("--config", default="c2", choices=sorted(OTHER_CONFIGS) + ["c2"],
                    help="c2 (default) = the headline line, BASELINE configs[1].  c1 / c3 / c4 / c5 = the other BASELINE configs, each with its "
                         "own metric string (never the headline): same timed loop, whole-path rates only")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"], help="c4 only (c5 is bf16, the rest fp32)")
    ap.add_argument("--no-fc-split3-leg", action="store_true", help="c2 only: skip the auxiliary `fc_split3_aux` leg (the same model with MPN_FC_SPLIT3)")
    ap.add_argument("--fc-arith", default="fp32", choices=["fp32", "split3"],
                    help="c2 / c3 only.  split3 = fc6 / fc7 on the bf16 matrix pipe (both operands as exact three-plane bf16 splits, six products, fp32 accumulate: "
                         "include/mpn.h MPN_FC_SPLIT3) — an AUXILIARY line with its own metric string and dtype, never the headline")
    ap.add_argument("--rois", type=int, default=N_ROIS,
                    help="c2 only: another proposal count for the headline model (e.g. 2000, scripts/eval_fastrcnn_voc2007.sh) -> an AUXILIARY line with its "
                         "own metric string; the default (1000) is the headline")
    ap.add_argument("--mixed-sizes", action="store_true",
                    help="c2 only: after the headline legs, a sustained leg over a dataset-shaped stream of DIFFERENT image sizes (bench.MIXED_SIZES, getImages' "
                         "rescale on the device) -> the `mixed_sizes` object; never the headline")
    ap.add_argument("--no-power-sensitivity", action="store_true", help="c2 only: skip the `power_sensitivity` leg (fc6 alone, back to back; debug-flavour library)")
    ap.add_argument("--mode", default="throughput", choices=["throughput", "latency"],
                    help="throughput (default) = the headline metric, images sharded over the ranks.  latency = ONE image's proposals and classes "
                         "sharded over the ranks (mpn_frcnn_test_one_sharded); its own metric string, never the headline")
    ap.add_argument("--sustained-seconds", type=float, default=3.0,
                    help="after the timed region: the same host-fed step for at least this long (and >= 800 steps at the default), clock / power "
                         "sampled beside it -> the `sustained` object of the JSON line; 0 = skip")
    ap.add_argument("--emulate-world", type=int, default=8,
                    help="latency mode on one GPU: also time rank 0's share of a world of this size (no all-gather), as a projection")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    import multipathnet_amd
    from multipathnet_amd import models, parallel

    lib = multipathnet_amd.load()
    if os.environ.get("MPN_HOOKS"):  # A/B TIMING ONLY, debug flavour: MPN_FLAVOUR=debug MPN_HOOKS="tower_lanes=0" python bench.py ... (printed to stderr; never the product)
        for kv in os.environ["MPN_HOOKS"].split():
            k, v = kv.split("=")
            getattr(lib, "mpn_debug_set_" + k)(int(v))
        sys.stderr.write("bench.py: DEBUG-FLAVOUR HOOKS %s — an A/B run, not a product measurement\n" % os.environ["MPN_HOOKS"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    share_gpu = os.environ.get("MPN_BENCH_SHARE_GPU") == "1" and world > 1  # TEST ONLY: every rank on device 0, gather through gloo
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("rank %d: only %d HIP devices are visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world == 1 and os.environ.get("MPN_BENCH_AFFINITY") == "1":
        affinity = pin_to_gpu_numa_node(dev_index)
    else:
        affinity = pin_to_gpu_numa_node(dev_index) if (world > 1 and not share_gpu) else "not set (single rank)" if world == 1 else "not set (shared GPU)"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CPU-side group: rendezvous, barriers, the MAX of the elapsed time.  No RCCL communicator is created here — the only one
        # a rank holds is the C ABI's (mpn_comm), bootstrapped by broadcasting rank 0's 128-byte id through this group.
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # the data-path collective: mpn_gather_dets (pack kernel + ncclAllGather through the C ABI).  Should the direct RCCL
    # communicator fail to come up on some node, the 11-KB gather goes through a torch.distributed RCCL group created for
    # that purpose instead — the bench line says which one ran; all ranks agree on the choice.
    comm, comm_err, nccl_group = None, "", None
    force_fail = os.environ.get("MPN_BENCH_FORCE_COMM_FAIL") == "1"  # TEST ONLY: take the fallback branch below as if the C-ABI communicator had failed
    if force_fail:
        comm_err = "MPN_BENCH_FORCE_COMM_FAIL=1 (test: the C-ABI communicator was not tried)"
    elif share_gpu:
        comm_err = "MPN_BENCH_SHARE_GPU=1"
    else:
        try:
            comm = parallel.Comm.from_torch_distributed()
        except Exception as e:  # noqa: BLE001
            comm_err = str(e).splitlines()[-1][:200]
    if world > 1:
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
        if comm is None and (not share_gpu or force_fail):
            # the fallback: a torch.distributed group for the 11-KB record gather.  RCCL on real multi-GPU runs; under the one-GPU
            # share mode (RCCL refuses two ranks on one device) the SAME branch runs with a gloo group, so that its control flow —
            # every rank agreeing on the choice, group creation, the gather through torch.distributed — is exercised by a test
            nccl_group = dist.new_group(backend="gloo" if share_gpu else "nccl")
    elif comm is None:
        raise SystemExit("bench.py: " + comm_err)
    if world == 1:
        gather_via = "no gather at world 1 (mpn_comm_init_rank(NULL, 1, 0) creates no RCCL communicator; nothing is exchanged)"
    elif comm is not None:
        gather_via = "mpn_gather_dets (RCCL through the C ABI; the rank's only RCCL communicator)"
    elif share_gpu and nccl_group is None:
        gather_via = "gloo all_gather of the packed record — TEST MODE MPN_BENCH_SHARE_GPU=1: all ranks share device 0, NOT a multi-GPU measurement"
    else:
        gather_via = ("torch.distributed all_gather_into_tensor on a fallback group (%s); C-ABI communicator failed: %s"
                      % ("gloo — TEST MODE MPN_BENCH_SHARE_GPU=1: all ranks share device 0, NOT a multi-GPU measurement" if share_gpu else "RCCL", comm_err))
    if args.mode == "latency":
        return latency_mode(args, torch, dist, models, parallel, comm, rank, world, dev, share_gpu, affinity)

    # the mixed-size leg's pinned host inputs are staged NOW, like the headline's: (diagnostic: MPN_BENCH_MIXED_PIN_LATE=1 stages them when the leg starts)
    mixed_pin = [(torch.from_numpy(i).pin_memory(), torch.from_numpy(b).pin_memory()) for i, b in mixed_size_inputs()] if args.mixed_sizes else None
    other = None
    if args.config != "c2" or args.rois != N_ROIS or args.fc_arith != "fp32":
        other = _cfg_vgg_frcnn_n(models, args) if args.config == "c2" else OTHER_CONFIGS[args.config](models, args)
        P, net, n_rois_cfg = other["params"], other["net"], other["n_rois"]
        im_np, boxes_np = synthetic_inputs()
        boxes_np = more_boxes(boxes_np, n_rois_cfg)
    else:
        P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=N_CLASSES, seed=557)
        net = models.FastRCNN(P, max_h=H, max_w=W, max_rois=N_ROIS)
        im_np, boxes_np = synthetic_inputs()
        n_rois_cfg = N_ROIS
    # four DIFFERENT pinned (image, proposals) sets in rotation, offset by rank: no step uploads the bytes the previous step did,
    # and no two ranks upload the same image in the same step (variant 0 = the SURVEY inputs, which the roofline / CPU legs use)
    n_rot = 4
    rot = [(im_np, boxes_np)] + [synthetic_inputs(v) for v in range(1, n_rot)]
    if other is not None:
        rot = [(i, more_boxes(b, n_rois_cfg)) for i, b in rot]
    im_host = [torch.from_numpy(i).clone().pin_memory() for i, _ in rot]
    boxes_host = [torch.from_numpy(b).clone().pin_memory() for _, b in rot]
    im_dev, boxes_dev = torch.from_numpy(im_np).to(dev), torch.from_numpy(boxes_np).to(dev)
    top_cap = net._dets.size(0)
    gathered = [torch.empty((world, top_cap * 6 + 1), dtype=torch.float32, device=dev) for _ in range(2)]
    gstream = torch.cuda.Stream(device=dev)  # the gather never blocks the compute stream for longer than one image of drift
    main = torch.cuda.current_stream(dev)

    state = {"pending": None, "seq": 0}

    def gather(bufs):
        if world > 1 and bufs is not None:  # RCCL all-gather of the scored-box record only (~11 KB per rank)
            main.wait_stream(gstream)       # bounded drift: the previous gather has finished before its buffers are reused
            gstream.wait_stream(main)       # the record's rows are ordered on `main` by the pipelined call that just returned
            with torch.cuda.stream(gstream):
                if comm is not None:
                    comm.gather_dets(bufs[0], bufs[1], out=gathered[state["seq"] & 1])
                elif share_gpu:  # test mode: through a CPU group (gloo; the fallback group when the test forced one), record staged on the host
                    rec = parallel.pack_record(bufs[0], bufs[1], top_cap).cpu()
                    gathered[state["seq"] & 1].copy_(parallel.gather_detections(rec, group=nccl_group), non_blocking=True)
                else:
                    parallel.gather_detections(parallel.pack_record(bufs[0], bufs[1], top_cap), group=nccl_group, out=gathered[state["seq"] & 1])

    def make_step(host_fed):
        def step():
            # Tester:test loop form: image i's NMS/top-k tail runs on the pipeline's side stream and overlaps image
            # i+1's trunk; its detections are stream-ordered one call later, when they are gathered.
            b = (state["seq"] + rank) % n_rot
            cur = net.test_one_pipelined_host(im_host[b], boxes_host[b]) if host_fed else net.test_one_pipelined(im_dev, boxes_dev)
            gather(state["pending"])
            state["pending"] = cur
            state["seq"] += 1
        return step

    def drain():
        net.flush()
        gather(state["pending"])
        state["pending"] = None
        main.wait_stream(gstream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(step):
        for _ in range(args.warmup):
            step()
        drain()
        fence()
        gc_was = gc.isenabled()
        gc.disable()  # no collector pause inside a 70 ms timed region
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()  # every step's tail and gather completes inside the timed region
        fence()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        state["per_rank_s"] = [dt]
        if world > 1:
            each = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(each, torch.tensor([dt], dtype=torch.float64))
            state["per_rank_s"] = [float(t.item()) for t in each]
            dt = max(state["per_rank_s"])  # the MAX over ranks is the job's time
        return dt

    def sustained(step, per_step_s):
        """the same host-fed step for >= --sustained-seconds (>= 800 steps on the headline config), clock / power sampled beside it"""
        if args.sustained_seconds <= 0:
            return None
        n = int(np.ceil(args.sustained_seconds / per_step_s))
        if args.config == "c2" and args.sustained_seconds >= 3.0:
            n = max(n, 800)
        drain()
        fence()
        sampler = ClockSampler(dev_index).start()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        drain()
        fence()
        dts = time.perf_counter() - t0
        sampler.stop()
        if world > 1:
            tt = torch.tensor([dts], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt.item())
        out_s = {"steps": n, "seconds": round(dts, 3), "value": round(n * n_rois_cfg * world / dts, 1), "unit": "proposals/s",
                 "ms_per_step": round(dts / n * 1e3, 4), "what": "host-fed steps back to back right after the timed region (same step function, "
                 "same rotation of pinned inputs), MAX over ranks; clock / power of rank 0's GPU sampled during the leg"}
        out_s.update(sampler.summary())
        return out_s

    dt = timed(make_step(True))            # the metric: host image + boxes in, H2D inside the timed region
    per_rank = [args.steps * n_rois_cfg / t for t in state["per_rank_s"]]
    ranks_info = {"rccl_ranks": comm.rccl_ranks if comm is not None else 0,
                  "rccl_ranks_what": "ncclCommCount of the C-ABI communicator, cross-checked with ncclCommUserRank at init (mpn_comm_rccl_ranks); "
                                     "0 = no RCCL communicator (one rank, or the fallback / share-GPU test paths); -1 = this RCCL has no ncclCommCount: unverified",
                  "per_rank_proposals_per_s": {"min": round(min(per_rank), 1), "max": round(max(per_rank), 1),
                                               "all": [round(v, 1) for v in per_rank]}}
    dt_res = min(timed(make_step(False)) for _ in range(2))  # inputs already resident in HBM (auxiliary figure: best of two passes)
    value = args.steps * n_rois_cfg * world / dt
    value_res = args.steps * n_rois_cfg * world / dt_res
    sus = sustained(make_step(True), dt / args.steps)
    if sus is not None:
        sus["vs_timed_region"] = round(sus["value"] / value, 4)
    if other is not None:  # one of the widened configurations: its own metric string; the same objects the headline line carries
        # profiled leg: HIP events around every kernel group of the un-pipelined call (as the headline's roofline leg)
        net.set_profiling(True)
        net.get_profile(reset=True)
        for _ in range(args.steps):
            net.test_one_async(im_dev, boxes_dev)
        torch.cuda.synchronize()
        prof = net.get_profile(reset=True)
        net.set_profiling(False)
        if rank == 0:
            peak = 2500e12 if other["dtype"] == "bf16" else FP32_MFMA_PEAK
            pk = "bf16" if other["dtype"] == "bf16" else "fp32"
            tfl = value / world * (other["flops"] / n_rois_cfg) / 1e12
            groups = other["groups"]
            save = 1.0 - 1.0 / WINO_MUL_RATIO
            kernels = {}
            for tag, (ms, cnt) in prof.items():
                if not cnt:
                    continue
                per_image_ms = ms / args.steps
                k = {"ms_per_image": round(per_image_ms, 4), "launch_groups_per_image": cnt / args.steps}
                if tag in groups:
                    g = groups[tag]
                    ex = g["alg"] - g["wino"] * save
                    k["what"] = g["label"]
                    k["algorithmic_gflop"] = round(g["alg"] / 1e9, 2)
                    k["executed_gflop"] = round(ex / 1e9, 2)
                    k["executed_tflops"] = round(ex / (per_image_ms * 1e-3) / 1e12, 2)
                    k["executed_frac_of_%s_mfma_peak" % pk] = round(ex / (per_image_ms * 1e-3) / peak, 4)
                    k["algorithmic_equiv_frac_of_%s_mfma_peak" % pk] = round(g["alg"] / (per_image_ms * 1e-3) / peak, 4)
                kernels[tag] = k
            dom = max((t for t in kernels if t in groups), key=lambda t: kernels[t]["ms_per_image"])
            tot_alg = sum(g["alg"] for g in groups.values())
            tot_ex = sum(g["alg"] - g["wino"] * save for g in groups.values())
            traffic, traffic_src, tj = None, None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC passes cannot run inside the timed process: measured offline
            if os.path.exists(tpath):
                tj = json.load(open(tpath)).get(other["key"])
                if tj and "dominant" in tj:
                    traffic, traffic_src = tj["dominant"], tj.get("_source")
            out = {"metric": other["metric"], "value": round(value, 1), "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": other["dtype"], "data": "synthetic",
                   "config": {"workload": other["workload"] + "; host image + boxes uploaded inside the step",
                              "parallelism": "image-sharded over %d rank%s, all-gather of scored boxes only via %s; cpu affinity: %s"
                                             % (world, "" if world == 1 else "s", gather_via, affinity)},
                   "value_inputs_resident": round(value_res, 1),
                   "sustained": sus, "ranks": ranks_info,
                   "whole_path": {"algorithmic_gflop_per_image": round(tot_alg / 1e9, 2), "executed_gflop_per_image": round(tot_ex / 1e9, 2),
                                  "executed_frac_of_%s_mfma_peak" % pk: round(value / world * (tot_ex / n_rois_cfg) / peak, 4),
                                  "algorithmic_equiv_frac_of_%s_mfma_peak" % pk: round(value / world * (tot_alg / n_rois_cfg) / peak, 4),
                                  "convolution_only_algorithmic_frac": round(tfl * 1e12 / peak, 4)},
                   "roofline": {"bound": "mfma", "kernel": dom + ": " + groups[dom]["label"], "achieved": kernels[dom]["executed_tflops"],
                                "peak": peak / 1e12, "unit": "TFLOP/s", "frac": kernels[dom]["executed_frac_of_%s_mfma_peak" % pk],
                                "algorithmic_equiv_frac": kernels[dom]["algorithmic_equiv_frac_of_%s_mfma_peak" % pk],
                                "algorithmic_bytes_per_image": other.get("alg_bytes"),
                                "algorithmic_bytes_what": "the TENSOR-LEVEL floor for ONE image: every tensor of the path (activations in the config's dtype, the "
                                                          "ROI-pooled tensors included) written once and read once however many layers consume it, weights once per "
                                                          "launch; layerwise_bytes_per_image (graph configs) = every LAYER reading its input and writing its output once",
                                "layerwise_bytes_per_image": other.get("layerwise_bytes"),
                                "traffic_bytes_per_image_all_kernels": tj.get("all_kernels_bytes_per_image") if tj else None,
                                "traffic_over_algorithmic": round(tj["all_kernels_bytes_per_image"] / other["alg_bytes"], 2) if (tj and tj.get("all_kernels_bytes_per_image") and other.get("alg_bytes")) else None,
                                "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_kernel": traffic,
                                "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) of the group's dominant kernel by name, averaged over its launches",
                                "traffic_source": traffic_src,
                                "flops_per_image": kernels[dom]["executed_gflop"] * 1e9, "ms_per_image": kernels[dom]["ms_per_image"],
                                "how": "HIP events on the launch stream around each kernel group of the un-pipelined call, %d profiled steps after the timed "
                                       "region; achieved / frac count the FLOPs the matrix pipe EXECUTES (layers on the Winograd kernel: direct FLOPs / 2.25), "
                                       "algorithmic_equiv_frac the direct-convolution FLOPs" % args.steps},
                   "kernels": kernels}
            if world == 1 and not args.no_cpu_baseline and other.get("cpu_baseline"):
                out["cpu_baseline"] = other["cpu_baseline"](im_np, boxes_np)
            if "split3" in other["key"] and "fc6" in kernels:
                fc6_ms = kernels["fc6"]["ms_per_image"]
                out["fc6_split3"] = {"ms_per_image": fc6_ms, "what": "operand split (split3_planes_kernel) + gemm_c8_split3_kernel + splitk_reduce_kernel",
                                     "bf16_mfma_flops_per_image": 6 * groups["fc6"]["alg"],
                                     "frac_of_bf16_mfma_peak_2.5PF": round(6 * groups["fc6"]["alg"] / (fc6_ms * 1e-3) / 2500e12, 4),
                                     "note": "the group's executed_frac_of_fp32_mfma_peak above prices fp32-equivalent FLOPs against the fp32 pipe's peak and may exceed 1: "
                                             "the six bf16 products run on the bf16 pipe (16x the fp32 MFMA rate)"}
            print(json.dumps(out))
            sys.stdout.flush()
        if comm is not None:
            comm.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline leg: same number of steps with HIP events around every kernel group (un-pipelined, single stream)
    net.set_profiling(True)
    net.get_profile(reset=True)
    for _ in range(args.steps):
        net.test_one_async(im_dev, boxes_dev)
    torch.cuda.synchronize()
    prof = net.get_profile(reset=True)
    net.set_profiling(False)

    # ---- auxiliary legs of the headline config (never the metric): fc6's sensitivity to clock / operand, and a mixed-size stream
    power_sens = None
    if world == 1 and not args.no_power_sensitivity:
        power_sens = power_sensitivity(torch, models, P, im_dev, boxes_dev, dev_index, prof)
    # Auxiliary leg (round 6; never the metric): the SAME model and inputs with fc6 / fc7 on the bf16 matrix pipe as exact three-plane splits with fp32
    # accumulation (include/mpn.h MPN_FC_SPLIT3) — so that the driver's own record carries this figure measured on the driver's box.  Inputs resident
    # in HBM, the pipelined form, --steps steps after --warmup; a failure here never touches the headline.  `bench.py --fc-arith split3` is the full line.
    fc_split3_aux = None
    if world == 1 and not args.no_fc_split3_leg:
        try:
            net3 = models.FastRCNN(P, max_h=H, max_w=W, max_rois=N_ROIS, fc_arith=1)
            for _ in range(args.warmup):
                net3.test_one_pipelined(im_dev, boxes_dev)
            net3.flush(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                net3.test_one_pipelined(im_dev, boxes_dev)
            net3.flush(); torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            s3, _ = net3.detect(im_dev, boxes_dev)
            s1, _ = net.detect(im_dev, boxes_dev)
            fc_split3_aux = {"value": round(args.steps * N_ROIS / dt3, 1), "unit": "proposals/s", "ms_per_step": round(dt3 / args.steps * 1e3, 4),
                             "dtype": "f32 (fc6 / fc7: 3-plane bf16 split, 6 products, fp32 accumulate)", "inputs": "resident in HBM",
                             "max_abs_score_difference_to_the_fp32_mfma_pipeline": float((s3 - s1).abs().max().item()),
                             "note": "AUXILIARY, never the metric: the headline above is pure fp32 MFMA.  Gate and accuracy: DESIGN.md 4.2b, tools/r06_split3_gate.sh"}
            del net3
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            fc_split3_aux = {"error": "%s: %s" % (type(e).__name__, e)}
    mixed = None
    if args.mixed_sizes:
        mixed = mixed_sizes_leg(torch, dist, models, P, dev, dev_index, world, rank, max(args.sustained_seconds, 1.0),
                                pin=mixed_pin if os.environ.get("MPN_BENCH_MIXED_PIN_LATE") != "1" else None)

    if rank == 0:
        cf = conv_flops(models.VGG16_CFG, H, W)
        flops = {"conv_wino": sum(f for f, v in cf if v == "conv_wino"), "conv_direct": sum(f for f, v in cf if v == "conv_direct"),
                 "fc6": 2.0 * N_ROIS * 25088 * 4096, "fc7": 2.0 * N_ROIS * 4096 * 4096, "heads": 2.0 * N_ROIS * 4096 * 5 * N_CLASSES}
        executed = dict(flops)
        executed["conv_wino"] = flops["conv_wino"] / WINO_MUL_RATIO
        launches = {"conv_wino": sum(1 for f, v in cf if v == "conv_wino"), "conv_direct": sum(1 for f, v in cf if v == "conv_direct")}
        hbm_bytes = {"roi_pool": 4.0 * (512 * 38 * 63 + 5 * N_ROIS + N_ROIS * 512 * 49),
                     "conv_direct": 4.0 * (8 * 602 * 1002 + 64 * 600 * 1000) + 4.0 * 64 * 27}     # conv1_1: C8P image in, 64-channel map out
        kernels = {}
        for tag, (ms, cnt) in prof.items():
            if cnt:
                per_image_ms = ms / args.steps
                k = {"ms_per_image": round(per_image_ms, 4), "launches_per_image": cnt / args.steps}
                if tag in flops:
                    k["executed_tflops"] = round(executed[tag] / (per_image_ms * 1e-3) / 1e12, 2)
                    k["executed_frac_of_fp32_mfma_peak"] = round(executed[tag] / (per_image_ms * 1e-3) / FP32_MFMA_PEAK, 4)
                if tag in hbm_bytes:
                    k["algorithmic_GBps"] = round(hbm_bytes[tag] / (per_image_ms * 1e-3) / 1e9, 1)
                    k["frac_of_hbm_peak_8TBps"] = round(hbm_bytes[tag] / (per_image_ms * 1e-3) / 8.0e12, 4)
                kernels[tag] = k
        dom = max((t for t in kernels if t in flops), key=lambda t: kernels[t]["ms_per_image"])
        n_launch = launches.get(dom, 1)
        sec = kernels[dom]["ms_per_image"] * 1e-3
        achieved = executed[dom] / sec / 1e12
        total_flops = sum(f for f, _ in cf) + N_ROIS * 2.0 * (25088 * 4096 + 4096 * 4096 + 4096 * 105)
        total_exec = total_flops - flops["conv_wino"] + executed["conv_wino"]
        out = {
            "metric": "proposals/sec (1000 ROIs, 600x1000 img) VGG-16 Fast R-CNN",
            "value": round(value, 1), "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VGG-16 Fast R-CNN, 1 image 600x1000 x 1000 ROIs per GPU per step, 21 classes, NMS 0.3, top-100 (BASELINE configs[1]); "
                                   "host image + boxes uploaded inside the step (pinned, copy stream, three staging sets)",
                       "parallelism": "image-sharded over %d rank%s (one process per GPU), all-gather of scored boxes only via %s; cpu affinity: %s; "
                                      "4 different pinned (image, proposals) sets in rotation, offset by rank"
                                      % (world, "" if world == 1 else "s", gather_via, affinity)},
            "value_inputs_resident": round(value_res, 1), "ms_per_step_inputs_resident": round(dt_res / args.steps * 1e3, 4),
            "sustained": sus, "ranks": ranks_info,
            "whole_path": {"algorithmic_gflop_per_image": round(total_flops / 1e9, 2), "executed_gflop_per_image": round(total_exec / 1e9, 2),
                           "executed_frac_of_fp32_mfma_peak": round(value / world * (total_exec / N_ROIS) / FP32_MFMA_PEAK, 4),
                           "algorithmic_equiv_frac_of_fp32_mfma_peak": round(value / world * (total_flops / N_ROIS) / FP32_MFMA_PEAK, 4)},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / FP32_MFMA_PEAK, 4), "traffic": None,
                         "flops_per_launch": executed[dom] / n_launch,
                         "avg_launch_ms": round(kernels[dom]["ms_per_image"] / n_launch, 5),
                         "how": "HIP events on the launch stream around each kernel group (a layer's split-K reduce kernel included), "
                                "%d profiled steps after the timed region" % args.steps},
            "kernels": kernels,
        }
        if dom == "conv_wino":
            out["roofline"]["algorithm"] = ("Winograd F(2x2,3x3) in fp32: achieved / frac count the FLOPs the matrix pipe executes "
                                            "(direct-convolution FLOPs / 2.25); algorithmic_equiv_* is the direct-convolution-equivalent rate")
            out["roofline"]["algorithmic_equiv_tflops"] = round(achieved * WINO_MUL_RATIO, 2)
            out["roofline"]["algorithmic_equiv_frac"] = round(achieved * WINO_MUL_RATIO * 1e12 / FP32_MFMA_PEAK, 4)
            out["roofline"]["algorithmic_flops_per_launch"] = flops[dom] / n_launch
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC passes cannot run inside the timed process: measured offline
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if dom in tj:
                out["roofline"]["traffic"] = tj[dom]["bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)"
                out["roofline"]["traffic_source"] = tj.get("_source")
        if power_sens is not None:
            out["power_sensitivity"] = power_sens
        if fc_split3_aux is not None:
            out["fc_split3_aux"] = fc_split3_aux
        if mixed is not None:
            out["mixed_sizes"] = mixed
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, im_np, boxes_np, args.cpu_rois)
        print(json.dumps(out))
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
