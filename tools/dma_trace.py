"""s_memtime trace of conv2d_c8i_bf16_dma_kernel (wave 0 of block 0) in the ResNet-50 bf16 per-ROI head: per stage the time from the
stage's first MFMA to the pre-wait point (k-step 0 + everything interleaved), the vmcnt wait, the barrier wait, and k-step 1.
Usage: python tools/dma_trace.py [kh]   (kh = 3: the last 3x3 layer, 1: the last pointwise layer).  s_memtime counts shader cycles on gfx950."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import multipathnet_amd
from multipathnet_amd import models
import bench
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
kh = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, N = 600, 1000, 1000
R = models.synthetic_resnet_params(depth=50, n_classes=21, seed=3)
net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=N, bf16=True)
im_np, boxes_np = bench.synthetic_inputs()
dev = torch.device("cuda", 0)
im, boxes = torch.from_numpy(im_np).to(dev), torch.from_numpy(boxes_np).to(dev)
for _ in range(2):
    net.detect(im, boxes)
torch.cuda.synchronize()
buf = torch.zeros(8 + 96 * 4, dtype=torch.int64, device="cuda")
lib.mpn_debug_set_bf16_trace(C.c_void_p(buf.data_ptr()), kh)
net.detect(im, boxes)
torch.cuda.synchronize()
lib.mpn_debug_set_bf16_trace(None, 3)
b = buf.cpu().numpy()
t0, t_epi, t_end, nst = b[0], b[1], b[2], b[3]
st = b[8:].reshape(96, 4)
print("block 0 of the last kh=%d layer: %d stages; kernel-start->first stage %d cycles, K loop %d, epilogue %d" % (kh, nst, st[0][0] - t0, t_epi - st[0][0], t_end - t_epi))
rows = [r for r in st.tolist() if r[0] != 0]
k0 = [r[1] - r[0] for r in rows if r[1]]
wv = [r[2] - r[1] for r in rows if r[1]]
wb = [r[3] - r[2] for r in rows if r[1]]
k1 = [rows[i + 1][0] - rows[i][3] for i in range(len(rows) - 1) if rows[i][3]]
per = [rows[i + 1][0] - rows[i][0] for i in range(len(rows) - 1)]
f = lambda v: "mean %.1f min %d max %d" % (np.mean(v), min(v), max(v)) if len(v) else "-"
print(" stages traced %d; per stage cycles: total %s" % (len(rows), f(per)))
print("   k-step 0 (+ DMA issue, fragment reads): %s" % f(k0))
print("   vmcnt wait: %s" % f(wv))
print("   zero + barrier: %s" % f(wb))
print("   k-step 1 (+ next stage's first fragment reads): %s" % f(k1))
print("   first 12 stages [k0, wait, barrier]: " + " ".join("[%d %d %d]" % (r[1] - r[0], r[2] - r[1], r[3] - r[2]) for r in rows[:12] if r[1]))
