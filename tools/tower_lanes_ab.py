"""A/B of the two tower lanes (round 6; mpn_debug_set_tower_lanes, debug flavour): the pipelined loop of BASELINE configs[2] / [3] bf16 / [4]
with the towers of an image one after the other on the launch stream (lanes 0: rounds 2-5) and on two lanes (1), interleaved legs.
Usage: MPN_FLAVOUR=debug python tools/tower_lanes_ab.py [c3] [c4] [c5] [images per leg]"""
import os, sys, time
os.environ.setdefault("MPN_FLAVOUR", "debug")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import multipathnet_amd
from multipathnet_amd import models

lib = multipathnet_amd.load()
which = [a for a in sys.argv[1:] if a in ("c3", "c4", "c5")] or ["c3", "c4", "c5"]
K = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 12
dev = torch.device("cuda", 0)
for cfg in which:
    if cfg == "c5":
        N = 2000
        net = models.InceptionFRCNN(models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557), max_h=600, max_w=1000, max_rois=N, bf16=True)
        name = "configs[4] Inception-v3 MultiPathNet bf16, 2000 ROIs"
    elif cfg == "c4":
        N = 1000
        net = models.ResNetFRCNN(models.synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=557), max_h=600, max_w=1000, max_rois=N, bf16=True)
        name = "configs[3] ResNet-50 MultiPathNet bf16, 1000 ROIs"
    else:
        N = 1000
        net = models.MultiPathNet(models.synthetic_mpnet_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=81, n_integral=6, seed=557), max_h=600, max_w=1000, max_rois=N)
        name = "configs[2] VGG-16 MultiPathNet fp32, 1000 ROIs"
    im, boxes = bench.synthetic_inputs()
    rng = np.random.default_rng(556)
    while boxes.shape[0] < N:
        boxes = np.concatenate([boxes, boxes[rng.permutation(boxes.shape[0])] * np.float32(0.97) + np.float32(1.0)])
    boxes = np.clip(boxes[:N], 1, [1000, 600, 1000, 600]).astype(np.float32)
    im, boxes = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)

    def leg(lanes):
        lib.mpn_debug_set_tower_lanes(lanes * (1 if cfg == "c3" else 2))   # graph towers: the lanes are a debug-flavour experiment (2)
        for _ in range(3):
            net.test_one_pipelined(im, boxes)
        net.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            net.test_one_pipelined(im, boxes)
        net.flush(); torch.cuda.synchronize()
        lib.mpn_debug_set_tower_lanes(1)
        return (time.perf_counter() - t0) / K * 1e3

    legs = [(0, 0, "towers one after the other, each pools its own operand (rounds 2-5)"), (1, 0, "two tower lanes"), (1, 1, "two tower lanes + shared operand (what ships)")] \
        if cfg == "c3" else [(0, 0, "towers one after the other (what ships)"), (1, 0, "two tower lanes")]
    res = {l[:2]: [] for l in legs}
    for rep in range(3):
        for ln, sh, _ in legs:
            lib.mpn_debug_set_tower_share(sh)
            res[(ln, sh)].append(leg(ln))
    lib.mpn_debug_set_tower_share(1)
    base = min(res[legs[0][:2]])
    print("%s  (%d images per leg, 3 interleaved repeats, best of)" % (name, K))
    for ln, sh, what in legs:
        b = min(res[(ln, sh)])
        print("  lanes %d share %d  %-72s %7.3f ms / image  (%+.2f ms, %+.1f %%)   all: %s" % (ln, sh, what, b, b - base, (b - base) / base * 100, " ".join("%.3f" % x for x in res[(ln, sh)])))
    del net
    torch.cuda.empty_cache()
