"""Generic A/B of debug-flavour hooks on the pipelined loop of one BASELINE config: interleaved legs, best of 3.
Usage: MPN_FLAVOUR=debug python tools/hook_ab.py <c2|c3|c4|c5> [images per leg] leg [leg ...]     leg = name=value[,name=value...] | base
  e.g. tools/hook_ab.py c5 12 base pool_exp=1 pool_exp=2 pool_exp=3"""
import os, sys, time
os.environ.setdefault("MPN_FLAVOUR", "debug")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import multipathnet_amd
from multipathnet_amd import models

lib = multipathnet_amd.load()
cfg = sys.argv[1]
args = sys.argv[2:]
K = 12
if args and args[0].isdigit():
    K = int(args.pop(0))
legs = args or ["base"]
dev = torch.device("cuda", 0)
if cfg == "c5":
    N = 2000
    net = models.InceptionFRCNN(models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557), max_h=600, max_w=1000, max_rois=N, bf16=True)
    name = "configs[4] Inception-v3 MultiPathNet bf16, 2000 ROIs"
elif cfg == "c4":
    N = 1000
    net = models.ResNetFRCNN(models.synthetic_resnet_mpn_params(depth=50, n_classes=81, n_integral=6, seed=557), max_h=600, max_w=1000, max_rois=N, bf16=True)
    name = "configs[3] ResNet-50 MultiPathNet bf16, 1000 ROIs"
elif cfg == "c2":
    N = 1000
    net = models.FastRCNN(models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557), max_h=600, max_w=1000, max_rois=N)
    name = "configs[1] VGG-16 Fast R-CNN fp32, 1000 ROIs (the headline)"
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
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import HOOK_DEFAULTS
DEFAULTS = dict(HOOK_DEFAULTS, pool_exp=0, tower_knock=0, mpn_pool_knock=0)


def run(leg):
    kv = [] if leg == "base" else [x.split("=") for x in leg.split(",")]
    for k, v in kv:
        getattr(lib, "mpn_debug_set_" + k)(int(v))
    for _ in range(3):
        net.test_one_pipelined(im, boxes)
    net.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        net.test_one_pipelined(im, boxes)
    net.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    for k, _ in kv:
        getattr(lib, "mpn_debug_set_" + k)(DEFAULTS[k])
    return dt


res = {l: [] for l in legs}
for rep in range(3):
    for l in legs:
        res[l].append(run(l))
base = min(res[legs[0]])
print("%s  (%d images per leg, 3 interleaved repeats, best of)" % (name, K))
for l in legs:
    b = min(res[l])
    print("  %-36s %7.3f ms / image  (%+.3f ms, %+.2f %%)   all: %s" % (l, b, b - base, (b - base) / base * 100, " ".join("%.3f" % x for x in res[l])))
