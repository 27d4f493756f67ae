"""BASELINE configs[4] backbone on ONE GPU: Inception-v3 Fast R-CNN (models/inceptionv3.lua graph), 600x1000 image, bf16 by default,
2000 ROIs — timing of the full per-image path (not a bench.py line; parity is tests/test_gpu_inception.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multipathnet_amd import models
import multipathnet_amd
if os.environ.get("MPN_HOOKS"):  # A/B on the debug flavour: MPN_FLAVOUR=debug MPN_HOOKS="bf16_bdir=0 bf16_dma_tn=256" python tools/...
    for kv in os.environ["MPN_HOOKS"].split():
        k, v = kv.split("=")
        getattr(multipathnet_amd.load(), "mpn_debug_set_" + k)(int(v))
if os.environ.get("MPN_SPLIT_MAX_TILES"):  # A/B: split-K only layers with fewer 128 x 128 tiles than this
    multipathnet_amd.load().mpn_debug_set_split_max_tiles(int(os.environ["MPN_SPLIT_MAX_TILES"]))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bf16 = "fp32" not in sys.argv[2:]
mpn = "mpn" in sys.argv[2:]  # BASELINE configs[4]: MultiPathNet towers (4 Foveal scales + box tower) on this backbone
G = models.synthetic_inception_mpn_params(n_classes=81, n_integral=6, seed=557) if mpn else models.synthetic_inception_v3_params(n_classes=81, seed=557)
net = models.InceptionFRCNN(G, max_h=600, max_w=1000, max_rois=N, bf16=bf16)
im, boxes = bench.synthetic_inputs()
rng = np.random.default_rng(556)
while boxes.shape[0] < N:  # more proposals of the same distribution
    boxes = np.concatenate([boxes, boxes[rng.permutation(boxes.shape[0])] * np.float32(0.97) + np.float32(1.0)])
boxes = np.clip(boxes[:N], 1, [1000, 600, 1000, 600]).astype(np.float32)
dev = torch.device("cuda", 0)
im, boxes = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
for _ in range(2):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
K = 3
t0 = time.perf_counter()
for _ in range(K):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K

def flops(ops, h, w):
    dims, f = {0: (h, w)}, 0.0
    for o in ops:
        sh_, sw_ = dims[o["src"]]
        oh, ow = (sh_ + 2 * o["ph"] - o["kh"]) // o["sh"] + 1, (sw_ + 2 * o["pw"] - o["kw"]) // o["sw"] + 1
        dims.setdefault(o["dst"], (oh, ow))
        if o["kind"] == 0:
            f += 2.0 * o["cout"] * o["cin"] * o["kh"] * o["kw"] * oh * ow
    return f, dims
ft, dims = flops(G["trunk_ops"], 600, 1000)
fh, _ = flops(G["head_ops"], 17, 17)
total = ft + N * fh * (len(G["head_towers"]) if mpn else 1)
net.set_profiling(True); net.get_profile(True)
net.test_one_async(im, boxes); torch.cuda.synchronize()
prof = net.get_profile(True)
peak = 2500e12 if bf16 else 157.3e12
print(("Inception-v3 MultiPathNet (5 towers, K=6)" if mpn else "Inception-v3 Fast R-CNN") + " %s, %d ROIs, 81 classes: %.2f ms/image  %.0f proposals/s  %.2f TFLOP/image  %.1f TFLOP/s (%.1f%% of the dtype's dense MFMA peak); feature map %dx%d" % (
    "bf16" if bf16 else "fp32", N, dt * 1e3, N / dt, total / 1e12, total / dt / 1e12, total / dt / peak * 100, *dims[G["feat_tensor"]]))
for k, (ms, n) in prof.items():
    if n: print("  %-12s %8.3f ms (%d launch groups)%s" % (k, ms, n, {"conv_direct": "  = trunk (stem + Mixed_5b..6e)", "fc6": "  = ROI pool + per-ROI Mixed_7a..7c + avgpool"}.get(k, "")))
print("n dets", int(net._n_dets.item()))
