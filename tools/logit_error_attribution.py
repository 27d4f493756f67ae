"""CPU-only measurement behind tests/test_gpu_fullsize.py::test_fullsize_trained_scale_logits_deltas_vs_oracle: how far two fp32 CPU
implementations of the same network (PyTorch-CPU oneDNN, the oracle plain C) are from each other and from a float64 head at the
trained / saturated score scales.  No device involved.  python tools/logit_error_attribution.py (about 2 min on 8 cores)."""
import os, sys, time, numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from multipathnet_amd import models
from oracle import mpn_oracle as O
from conftest import saturated_heads
O.build()
P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557)
Pn={k: ([t.numpy() for t in v] if isinstance(v, list) and v and hasattr(v[0], "numpy") else (v.numpy() if hasattr(v, "numpy") else v)) for k, v in P.items()}
im, boxes = bench.synthetic_inputs()
t=time.time()
feat_o = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"])
print("oracle trunk", time.time()-t)
with torch.no_grad():
    x = torch.from_numpy(O.image_transform(im, **O.ROSS)).unsqueeze(0)
    li=0
    for item in models.VGG16_CFG:
        if item=="P": x=F.max_pool2d(x,2,2,ceil_mode=True)
        else:
            x=F.relu(F.conv2d(x,P["conv_w"][li],P["conv_b"][li],padding=1)); li+=1
    feat_t=x[0].numpy()
print("conv5 torch vs oracle: max abs", np.abs(feat_t-feat_o).max(), "scale", np.abs(feat_o).max())
rois=O.project_im_rois(boxes,1.0)
pooled_o,_=O.roi_pool(feat_o,rois,7,7,1/16.)
pooled_t,_=O.roi_pool(feat_t,rois,7,7,1/16.)
def head32(pooled):
    with torch.no_grad():
        h=torch.from_numpy(pooled.reshape(1000,-1))
        h=F.relu(F.linear(h,P["fc6_w"],P["fc6_b"]))
        return F.relu(F.linear(h,P["fc7_w"],P["fc7_b"]))
def head64(pooled, idx):
    with torch.no_grad():
        h=torch.from_numpy(pooled.reshape(1000,-1)[idx]).double()
        h=F.relu(F.linear(h,P["fc6_w"].double(),P["fc6_b"].double()))
        return F.relu(F.linear(h,P["fc7_w"].double(),P["fc7_b"].double()))
fc7_t=head32(pooled_t)
Q=saturated_heads(P, fc7_t, boxes, 21)
idx=np.random.default_rng(7).choice(1000,100,replace=False)
Qn=dict(Pn); Qn.update({k:Q[k].numpy() for k in ("cls_w","cls_b","bbox_w","bbox_b")})
lo,_=O.frcnn_head(feat_o, rois[idx], Qn)               # oracle fp32 end to end
lt=F.linear(fc7_t[idx],Q["cls_w"],Q["cls_b"]).numpy()   # torch fp32 end to end
l64_o=F.linear(head64(pooled_o,idx),Q["cls_w"].double(),Q["cls_b"].double()).numpy()  # exact head on the oracle's conv5
l64_t=F.linear(head64(pooled_t,idx),Q["cls_w"].double(),Q["cls_b"].double()).numpy()  # exact head on torch's conv5
print("saturated: torch32 vs oracle32 logits", np.abs(lt-lo).max())
print("  oracle32 head error vs exact head (same conv5)", np.abs(lo-l64_o).max())
print("  torch32 head error vs exact head (same conv5)", np.abs(lt-l64_t).max())
print("  trunk difference (torch vs oracle conv5) through the exact head", np.abs(l64_t-l64_o).max())
T=models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=21, seed=557, head_scale="trained")
Tn=dict(Pn); Tn.update({k:T[k].numpy() for k in ("cls_w","cls_b","bbox_w","bbox_b")})
lo,_=O.frcnn_head(feat_o, rois[idx], Tn)
lt=F.linear(fc7_t[idx],T["cls_w"],T["cls_b"]).numpy()
l64_o=F.linear(head64(pooled_o,idx),T["cls_w"].double(),T["cls_b"].double()).numpy()
l64_t=F.linear(head64(pooled_t,idx),T["cls_w"].double(),T["cls_b"].double()).numpy()
print("trained: torch32 vs oracle32", np.abs(lt-lo).max(), "oracle head err", np.abs(lo-l64_o).max(), "torch head err", np.abs(lt-l64_t).max(), "trunk diff via exact head", np.abs(l64_t-l64_o).max())
