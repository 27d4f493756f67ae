"""Is a configuration bound by the host's launch rate or by the GPU?  Enqueues `steps` pipelined images without synchronising and
reports (a) host time per call to enqueue them, (b) wall time per image once the device has drained.  (a) ~ (b): the host is the
limit (launch overhead; ~45 launches per AlexNet image); (a) << (b): the GPU is.   usage: python tools/host_enqueue_probe.py [c1|c2] [steps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from multipathnet_amd import models  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
H, W = 600, 1000
rng = np.random.default_rng(1)
if cfg == "c1":
    N = 300
    net = models.AlexNetFRCNN(models.synthetic_alexnet_params(n_classes=21, seed=557), max_h=H, max_w=W, max_rois=N)
else:
    N = 1000
    net = models.FastRCNN(models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557), max_h=H, max_w=W, max_rois=N)
im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
c = rng.uniform([1, 1], [W, H], (N, 2))
wh = np.exp(rng.uniform(np.log(16), np.log(400), (N, 2)))
boxes = torch.from_numpy(np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)).to(dev)
for _ in range(10):
    net.test_one_pipelined(im, boxes)
net.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    net.test_one_pipelined(im, boxes)
t1 = time.perf_counter()
net.flush()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s resident: host enqueue %.1f us / image, wall %.1f us / image (%d images)" % (cfg, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6, steps))
# host-fed form: pinned image + boxes uploaded on the pipeline's copy stream inside the call
imh = [im.cpu().clone().pin_memory() for _ in range(4)]
bxh = [boxes.cpu().clone().pin_memory() for _ in range(4)]
for i in range(10):
    net.test_one_pipelined_host(imh[i & 3], bxh[i & 3])
net.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    net.test_one_pipelined_host(imh[i & 3], bxh[i & 3])
t1 = time.perf_counter()
net.flush()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s host-fed: host enqueue %.1f us / image, wall %.1f us / image" % (cfg, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
# the upload alone
cs = torch.cuda.Stream()
d = torch.empty_like(im)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(cs):
    for i in range(50):
        d.copy_(imh[i & 3], non_blocking=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("upload alone: host %.1f us / copy, wall %.1f us / copy of %.1f MB" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6, im.numel() * 4 / 1e6))
# resident pipeline with free-running uploads on another stream (no dependency between the two): copy / compute contention alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    net.test_one_pipelined(im, boxes)
    with torch.cuda.stream(cs):
        d.copy_(imh[i & 3], non_blocking=True)
net.flush()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s resident + independent uploads: wall %.1f us / image" % (cfg, (t2 - t0) / steps * 1e6))
