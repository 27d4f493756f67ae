import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, multipathnet_amd
from multipathnet_amd import models
multipathnet_amd.load()
dev = torch.device("cuda", 0)
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
net = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=bench.N_ROIS, scale=600, max_size=1000)
pin = [(torch.from_numpy(i).pin_memory(), torch.from_numpy(b).pin_memory()) for i, b in bench.mixed_size_inputs()]
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        net.test_one_pipelined_host(*pin[k % 6])
    net.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(12)
for n in (6, 60, 120, 300, 600):
    print("rotation, %3d steps: %.3f ms per image" % (n, run(n)))
s = bench.ClockSampler(0).start()
print("with the clock sampler thread: %.3f ms per image" % run(600))
s.stop(); print(s.summary())
# host time per call
t0 = time.perf_counter()
for k in range(60):
    net.test_one_pipelined_host(*pin[k % 6])
t1 = time.perf_counter(); net.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.3f ms per call, drain %.3f ms" % ((t1 - t0) / 60 * 1e3, (t2 - t1) * 1e3))
# a SECOND pipeline handle alive on the same device (as in bench.py, where the headline handle exists beside the mixed-size one): each handle
# owns a copy stream and a high-priority side stream; HIP multiplexes a process's streams onto a few hardware queues
other = models.FastRCNN(P, max_h=600, max_w=1000, max_rois=bench.N_ROIS)
im0, bx0 = bench.synthetic_inputs()
pin0 = (torch.from_numpy(im0).pin_memory(), torch.from_numpy(bx0).pin_memory())
for _ in range(8):
    other.test_one_pipelined_host(*pin0)
other.flush(); torch.cuda.synchronize()
print("rotation with a second (idle, used once) handle alive: %.3f ms per image" % run(300))
other.close() if hasattr(other, "close") else None
del other
torch.cuda.synchronize()
print("rotation after the second handle was destroyed: %.3f ms per image" % run(300))
