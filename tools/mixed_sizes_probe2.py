import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, multipathnet_amd
from multipathnet_amd import models
import torch.distributed as dist
multipathnet_amd.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
out = bench.mixed_sizes_leg(torch, dist, models, P, dev, 0, 1, 0, 1.0)
print("leg alone:", out["ms_per_image"], out["power_w"])
net = models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
im0, bx0 = bench.synthetic_inputs()
pin0 = (torch.from_numpy(im0).pin_memory(), torch.from_numpy(bx0).pin_memory())
imd, bxd = torch.from_numpy(im0).to(dev), torch.from_numpy(bx0).to(dev)
for _ in range(30):
    net.test_one_pipelined_host(*pin0)
net.flush(); torch.cuda.synchronize()
out = bench.mixed_sizes_leg(torch, dist, models, P, dev, 0, 1, 0, 1.0)
print("leg after a host-fed headline loop:", out["ms_per_image"], out["power_w"])
for _ in range(30):
    net.test_one_pipelined(imd, bxd)
net.flush(); torch.cuda.synchronize()
out = bench.mixed_sizes_leg(torch, dist, models, P, dev, 0, 1, 0, 1.0)
print("leg after a device-fed loop:", out["ms_per_image"], out["power_w"])
net.set_profiling(True); net.get_profile(reset=True)
for _ in range(10):
    net.test_one_async(imd, bxd)
torch.cuda.synchronize(); net.get_profile(reset=True); net.set_profiling(False)
out = bench.mixed_sizes_leg(torch, dist, models, P, dev, 0, 1, 0, 1.0)
print("leg after the profiled leg:", out["ms_per_image"], out["power_w"])
