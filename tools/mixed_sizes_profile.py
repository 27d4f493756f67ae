"""Where a mixed-size stream's time goes (bench.MIXED_SIZES through getImages' rescale on the device): per size, the un-pipelined
per-group HIP-event times when the size repeats, and the extra a size CHANGE costs.  python tools/mixed_sizes_profile.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, multipathnet_amd
from multipathnet_amd import models
multipathnet_amd.load()
dev = torch.device("cuda", 0)
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
net = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=bench.N_ROIS, scale=600, max_size=1000)
stream = [(torch.from_numpy(i).to(dev), torch.from_numpy(b).to(dev)) for i, b in bench.mixed_size_inputs()]
pin = [(torch.from_numpy(i).pin_memory(), torch.from_numpy(b).pin_memory()) for i, b in bench.mixed_size_inputs()]

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    net.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(2):
    for im, bx in stream:
        net.test_one_async(im, bx)
rep, alt = [], []
for i, (im, bx) in enumerate(stream):
    ms_same = timed(lambda k: net.test_one_async(im, bx), 12)                      # the same size back to back, device-resident, un-pipelined
    net.set_profiling(True); net.get_profile(reset=True)
    for _ in range(6):
        net.test_one_async(im, bx)
    torch.cuda.synchronize()
    prof = net.get_profile(reset=True); net.set_profiling(False)
    o = stream[(i + 1) % len(stream)]
    ms_alt = timed(lambda k: net.test_one_async(*(o if k & 1 else (im, bx))), 12)  # alternating with the next size: every call changes the size
    h, w, n = bench.MIXED_SIZES[i]
    rep.append(ms_same)
    print("%4dx%-4d %4d ROIs: same size %6.3f ms | groups %s" % (h, w, n, ms_same, "  ".join("%s %.3f" % (k, v[0] / 6) for k, v in prof.items() if v[1])))
    alt.append(ms_alt)
for i in range(len(stream)):
    j = (i + 1) % len(stream)
    print("alternating %d <-> %d: %6.3f ms per image vs %6.3f for the two sizes repeated: +%.3f ms per size change" % (i, j, alt[i], (rep[i] + rep[j]) / 2, alt[i] - (rep[i] + rep[j]) / 2))
ms_pipe = timed(lambda k: net.test_one_pipelined_host(*pin[k % len(pin)]), 60)
ms_pipe_same = np.mean([timed(lambda k: net.test_one_pipelined_host(*pin[i]), 12) for i in range(len(pin))])
print("host-fed pipelined: rotation %.3f ms per image; each size repeated (mean of the six) %.3f ms" % (ms_pipe, ms_pipe_same))
