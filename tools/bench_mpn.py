"""BASELINE configs[2]: VGG-16 MultiPathNet (4 foveal towers + het tower, skip concat, K=6 integral classifiers, C=81),
1000 ROIs, 600x1000 image — timing of the full per-image path (not a bench.py line; parity is tests/test_gpu_pipeline.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multipathnet_amd import models
P = models.synthetic_mpnet_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=81, n_integral=6, seed=557)
net = models.MultiPathNet(P, max_h=600, max_w=1000, max_rois=1000)
im, boxes = bench.synthetic_inputs()
dev = torch.device("cuda", 0)
im, boxes = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
for _ in range(2):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
net.set_profiling(True); net.get_profile(True)
net.test_one_async(im, boxes); torch.cuda.synchronize()
prof = net.get_profile(True)
flops = 367.74e9 + 5 * 1000 * 2 * (25088 * 4096 + 4096 * 4096) + 49 * 1000 * 2 * 512 * (1280 + 1024 + 1024 + 512 + 1280) + 1000 * 2 * (16384 * 486 + 4096 * 324)
print("MultiPathNet VGG-16 C3: %.2f ms/image  %.0f proposals/s  %.1f TFLOP/s (%.1f%% of fp32 MFMA peak)" % (dt * 1e3, 1000 / dt, flops / dt / 1e12, flops / dt / 157.3e12 * 100))
for k, (ms, n) in prof.items():
    if n: print("  %-12s %8.3f ms (%d launch groups)" % (k, ms, n))
scores, bbox = net.detect(im, boxes)
print("scores row sums", float(scores.sum(1).min()), float(scores.sum(1).max()), "n dets", int(net._n_dets.item()))
