"""BASELINE configs[3] shape on ONE GPU: ResNet-50 Fast R-CNN (models/resnet.lua graph, fp32), 1000 ROIs, 600x1000 image —
timing of the full per-image path (not a bench.py line; parity is tests/test_gpu_resnet.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multipathnet_amd import models
import multipathnet_amd
if os.environ.get("MPN_HOOKS"):  # A/B on the debug flavour: MPN_FLAVOUR=debug MPN_HOOKS="bf16_bdir=0 bf16_dma_tn=256" python tools/...
    for kv in os.environ["MPN_HOOKS"].split():
        k, v = kv.split("=")
        getattr(multipathnet_amd.load(), "mpn_debug_set_" + k)(int(v))
if os.environ.get("MPN_BF16_DMA"):  # A/B: 0 = never use the LDS-DMA bf16 convolution, 1 = large layers (default), 2 = every eligible layer
    multipathnet_amd.load().mpn_debug_set_bf16_dma(int(os.environ["MPN_BF16_DMA"]))
if os.environ.get("MPN_DMA_TN"):  # A/B: pixel-tile width of the bf16 LDS-DMA convolution kernel (128 / 256; 0 = per layer)
    multipathnet_amd.load().mpn_debug_set_bf16_dma_tn(int(os.environ["MPN_DMA_TN"]))
if os.environ.get("MPN_SPLIT_TARGET"):  # A/B: split-K block target of the bf16 128 x 128 convolution kernel (0 = never split)
    multipathnet_amd.load().mpn_debug_set_bf16_split_target(int(os.environ["MPN_SPLIT_TARGET"]))
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mpn = "mpn" in sys.argv[3:]
bf16 = "bf16" in sys.argv[3:]  # BASELINE configs[3]: MultiPathNet on the ResNet backbone (this library's extension)
R = models.synthetic_resnet_mpn_params(depth=depth, n_classes=81, n_integral=6, seed=557) if mpn else models.synthetic_resnet_params(depth=depth, n_classes=21, seed=557)
net = models.ResNetFRCNN(R, max_h=600, max_w=1000, max_rois=N, bf16=bf16)
im, boxes = bench.synthetic_inputs()
dev = torch.device("cuda", 0)
im, boxes = torch.from_numpy(im).to(dev), torch.from_numpy(boxes[:N]).to(dev)
for _ in range(2):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
K = 3
t0 = time.perf_counter()
for _ in range(K):
    net.test_one_pipelined(im, boxes)
net.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
# algorithmic FLOPs: every convolution 2*Cout*Cin*k*k*OH*OW (trunk on the image, head per ROI)
def conv_flops(blocks, h, w, cin0):
    f, cin = 0.0, cin0
    for b in blocks:
        bh, bw = h, w
        if b["shortcut"] is not None:
            ws, _, st = b["shortcut"]
            f += 2.0 * ws.shape[0] * ws.shape[1] * ((h - 1) // st + 1) * ((w - 1) // st + 1)
        for (wt, _, st, pd) in b["convs"]:
            k = wt.shape[2]
            bh, bw = (bh + 2 * pd - k) // st + 1, (bw + 2 * pd - k) // st + 1
            f += 2.0 * wt.shape[0] * wt.shape[1] * k * k * bh * bw
        h, w = bh, bw
    return f, h, w
h1, w1 = (600 + 6 - 7) // 2 + 1, (1000 + 6 - 7) // 2 + 1
f_trunk = 2.0 * 64 * 3 * 49 * h1 * w1
h2, w2 = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1
ft, fh_, fw_ = conv_flops(R["trunk_blocks"], h2, w2, 64)
fhead, _, _ = conv_flops(R["head_blocks"], 14, 14, 0)
flops = f_trunk + ft + N * fhead * (len(R["head_towers"]) if mpn else 1)
net.set_profiling(True); net.get_profile(True)
net.test_one_async(im, boxes); torch.cuda.synchronize()
prof = net.get_profile(True)
peak = 2500e12 if bf16 else 157.3e12
print(("ResNet-%d MultiPathNet (5 towers, K=6, 81 classes)" if mpn else "ResNet-%d Fast R-CNN") % depth + (" bf16" if bf16 else " fp32") + ", %d ROIs: %.2f ms/image  %.0f proposals/s  %.2f TFLOP/image  %.1f TFLOP/s (%.1f%% of the dtype's dense MFMA peak); feature map %dx%d" % (
    N, dt * 1e3, N / dt, flops / 1e12, flops / dt / 1e12, flops / dt / peak * 100, fh_, fw_))
for k, (ms, n) in prof.items():
    if n: print("  %-12s %8.3f ms (%d launch groups)%s" % (k, ms, n, {"conv_direct": "  = ResNet trunk", "fc6": "  = ROI pool + per-ROI layer4 + avgpool"}.get(k, "")))
print("n dets", int(net._n_dets.item()))
