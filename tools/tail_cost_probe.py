"""What does image i's select -> NMS -> top-k tail cost image i+1's trunk in the pipelined form?  The headline loop (host-fed pipelined
test_one, four pinned inputs in rotation) with (a) the product's tail, (b) a score threshold nothing passes (the tail's kernels run on empty
tables), (c) the fused NMS kernel forced under the trunk.  Debug flavour for all three (the knobs live there).
python tools/tail_cost_probe.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from multipathnet_amd import models, _lib

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dev = torch.device("cuda:0")
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
rot = [bench.synthetic_inputs(v) for v in range(4)]
im_host = [torch.from_numpy(i).clone().pin_memory() for i, _ in rot]
boxes_host = [torch.from_numpy(b).clone().pin_memory() for _, b in rot]
lib = _lib.load("debug")


def run(net, label):
    def loop(n):
        for k in range(n):
            net.test_one_pipelined_host(im_host[k % 4], boxes_host[k % 4])
        net.flush()
        torch.cuda.synchronize()
    loop(12)
    t0 = time.perf_counter(); loop(40); per = (time.perf_counter() - t0) / 40
    n = max(40, int(secs / per))
    t0 = time.perf_counter(); loop(n); per = (time.perf_counter() - t0) / n
    print("%-58s %.4f ms / image  = %.1f k proposals/s  (%d images)" % (label, per * 1e3, bench.N_ROIS / per / 1e3, n))
    return per


with _lib.debug_hooks():
    lib.mpn_debug_set_defer_heads(0)
    r4 = run(models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS), "heads / decode / select on the launch stream (as before)")
    lib.mpn_debug_set_defer_heads(1)
    a = run(models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS), "product tail (chain under the trunk)")
    b = run(models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS, score_thresh=10.0), "nothing passes the score threshold (empty tables)")
    lib.mpn_debug_set_nms_fused(2)
    c = run(models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS), "fused NMS kernel forced under the trunk")
    lib.mpn_debug_set_nms_fused(1)
    a2 = run(models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS), "product tail again")
print("handing the heads over to the side stream after fc7: %.1f us per image" % ((r4 - min(a, a2)) * 1e6))
print("the tail costs the pipelined loop %.1f us per image (%.2f %%); the fused kernel there %.1f us more" % ((min(a, a2) - b) * 1e6, (min(a, a2) - b) / min(a, a2) * 100, (c - min(a, a2)) * 1e6))
