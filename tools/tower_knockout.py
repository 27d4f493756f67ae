"""VERDICT r5 task 1, step 1: the CEILING of a first per-ROI layer that pools its ROIs on chip instead of reading a materialised pooled tensor
(models/inceptionv3.lua:27-43, models/resnet.lua:28-50), measured BEFORE building it — the way round 5 priced the Winograd prologue.
Debug flavour (mpn_debug_set_tower_knock; timing only, the knocked-out runs compute garbage):
  knock 0  the pipeline as shipped
  knock 1  the ROI pooling launches of the towers skipped (their 0.9-GB / 0.4-GB tensor is never written)
  knock 3  + the convolutions that read the pooled tensor fetch every pixel fragment from ONE L1-resident 1-KiB window — no kernel can get its
           operand cheaper, whatever it pools from
Usage: MPN_FLAVOUR=debug python tools/tower_knockout.py [c5|c4] [images per leg]"""
import os, sys, time
os.environ.setdefault("MPN_FLAVOUR", "debug")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import multipathnet_amd
from multipathnet_amd import models

lib = multipathnet_amd.load()
which = [a for a in sys.argv[1:] if a in ("c5", "c4")] or ["c5", "c4"]
K = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 12
dev = torch.device("cuda", 0)
for cfg in which:
    if cfg == "c5":
        N = 2000
        G = models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557)
        net = models.InceptionFRCNN(G, max_h=600, max_w=1000, max_rois=N, bf16=True)
        name = "configs[4] Inception-v3 MultiPathNet bf16, 2000 ROIs"
    else:
        N = 1000
        R = models.synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=557)
        net = models.ResNetFRCNN(R, max_h=600, max_w=1000, max_rois=N, bf16=True)
        name = "configs[3] ResNet-50 MultiPathNet bf16, 1000 ROIs"
    im, boxes = bench.synthetic_inputs()
    rng = np.random.default_rng(556)
    while boxes.shape[0] < N:
        boxes = np.concatenate([boxes, boxes[rng.permutation(boxes.shape[0])] * np.float32(0.97) + np.float32(1.0)])
    boxes = np.clip(boxes[:N], 1, [1000, 600, 1000, 600]).astype(np.float32)
    im, boxes = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)

    def leg(knock):
        lib.mpn_debug_set_tower_knock(knock)
        for _ in range(3):
            net.test_one_pipelined(im, boxes)
        net.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            net.test_one_pipelined(im, boxes)
        net.flush(); torch.cuda.synchronize()
        lib.mpn_debug_set_tower_knock(0)
        return (time.perf_counter() - t0) / K * 1e3

    res = {0: [], 1: [], 3: []}
    for rep in range(3):            # interleaved legs: clock / box drift lands on all three alike
        for k in (0, 1, 3):
            res[k].append(leg(k))
    base = min(res[0])
    print("%s  (%d images per leg, 3 interleaved repeats, best of)" % (name, K))
    for k, what in ((0, "as shipped"), (1, "ROI pooling launches skipped"), (3, "+ first-layer pixel fragments from one L1-resident window")):
        b = min(res[k])
        print("  knock %d  %-58s %7.3f ms / image  (%+.2f ms, %+.1f %%)   all: %s" % (k, what, b, b - base, (b - base) / base * 100, " ".join("%.3f" % x for x in res[k])))
    del net
    torch.cuda.empty_cache()
