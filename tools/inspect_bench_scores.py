import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multipathnet_amd import models
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557)
net = models.FastRCNN(P, max_h=600, max_w=1000, max_rois=1000)
im, boxes = bench.synthetic_inputs()
dev = torch.device("cuda", 0)
s, b = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
s = s.cpu().numpy(); b = b.cpu().numpy()
print("score range", s.min(), s.max(), "row0", s[0, :5])
for j in (1, 2, 20):
    u, c = np.unique(s[:, j], return_counts=True)
    print("class", j, "unique", len(u), "max run", c.max(), "n runs>1", (c > 1).sum())
net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
keep, idx, nk = net.nms_results()
print("n_keep", nk.cpu().numpy())
feat = net.debug_tensor("conv5", (512, 38, 63)).cpu().numpy()
print("conv5 stats mean %.4g max %.4g frac zero %.3f" % (feat.mean(), feat.max(), (feat == 0).mean()))
fc7 = net.debug_tensor("fc7", (1000, 4096)).cpu().numpy()
print("fc7 mean %.4g max %.4g frac zero %.3f; identical rows to row0: %d" % (fc7.mean(), fc7.max(), (fc7 == 0).mean(), (np.abs(fc7 - fc7[0]).max(1) == 0).sum()))
