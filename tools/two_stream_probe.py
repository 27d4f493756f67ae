"""Probe: does keeping TWO images in flight on one GPU (two pipeline handles, two HIP streams, images alternating) raise the
throughput of the per-image path?  Layer tails (a launch of B blocks on 256 one-block CUs leaves CUs idle in its last round), launch
gaps and the small latency-bound kernels of one image can then be filled by the other image's kernels.
    python tools/two_stream_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multipathnet_amd import models

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
nets = [models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS) for _ in range(2)]
im_np, boxes_np = bench.synthetic_inputs()
pins = [(torch.from_numpy(im_np).clone().pin_memory(), torch.from_numpy(boxes_np).clone().pin_memory()) for _ in range(4)]
# argv[2] = "prio": the second stream at LOWER priority, so that its image's blocks only fill what the first stream leaves idle
prio = len(sys.argv) > 2 and sys.argv[2] == "prio"
streams = [torch.cuda.Stream(device=dev, priority=-1 if prio else 0), torch.cuda.Stream(device=dev, priority=0)]


def run(n_streams, host_fed=True):
    def step(i):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            nets[k].test_one_pipelined_host(*pins[i % 4]) if host_fed else None
    for i in range(8):
        step(i)
    for k in range(n_streams):
        with torch.cuda.stream(streams[k]):
            nets[k].flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    for k in range(n_streams):
        with torch.cuda.stream(streams[k]):
            nets[k].flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3


for rep in range(2):
    a = run(1)
    b = run(2)
    print("one image in flight: %.4f ms/image (%.1f k proposals/s)   two in flight: %.4f ms/image (%.1f k proposals/s)   ratio %.3f"
          % (a, 1000 / a, b, 1000 / b, a / b))
