#!/usr/bin/env python
"""bench.py — proposals/sec of the per-image detection hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic 600x1000 image with 1000 ROIs through VGG-16
Fast R-CNN (BASELINE configs[1]): image transform -> 13 conv / 4 pool trunk -> ROI pool -> fc6/fc7 ->
cls/bbox heads -> softmax / BBoxNorm / decode / clamp -> per-class NMS -> top-100 -> (N>1) all-gather of
the scored-box record.  Inputs are resident in HBM before the timed region; weights are seeded
random (no pretrained blobs offline).  Images shard across ranks (weak scaling, one image per rank
per step); value = all ranks' proposals / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, fp32-MFMA bound; per-launch
durations from HIP events recorded on the launch stream in a second, equally long profiled pass) and
"cpu_baseline" (the oracle's CPU restatement timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, N_ROIS, N_CLASSES = 600, 1000, 1000, 21
FP32_MFMA_PEAK = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def conv_flops(cfg, h, w):
    """per-layer algorithmic FLOPs (2*H*W*Cin*9*Cout with the REAL Cin) and which tile variant runs it"""
    out, cin = [], 3
    for item in cfg:
        if item == "P":
            h, w = (h + 1) // 2, (w + 1) // 2
        else:
            out.append((2.0 * h * w * cin * 9 * item, "conv_wino" if cin >= 16 else "conv_direct"))
            cin = item
    return out


def synthetic_inputs():
    rng = np.random.default_rng(555)
    im = rng.random((3, H, W), dtype=np.float32)
    rng = np.random.default_rng(556)
    boxes = np.zeros((0, 4), np.float32)
    while boxes.shape[0] < N_ROIS:  # centre uniform, log-uniform w,h in [16,600], clipped, area > 2
        c = rng.uniform([1, 1], [W, H], (2 * N_ROIS, 2))
        wh = np.exp(rng.uniform(np.log(16), np.log(600), (2 * N_ROIS, 2)))
        b = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 1, [W, H, W, H]).astype(np.float32)
        b = b[(b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 2]
        boxes = np.concatenate([boxes, b])[:N_ROIS]
    return im, boxes


def cpu_baseline(P, im, boxes, rois_sample):
    """The oracle (CPU restatement, OpenMP over the host cores) on one image; the ROI head is timed on a
    bounded sample of ROIs and scaled (rows are independent), NMS on all classes of the sample."""
    from oracle import mpn_oracle as O
    Pn = {k: ([t.numpy() for t in v] if isinstance(v, list) and v and hasattr(v[0], "numpy") else (v.numpy() if hasattr(v, "numpy") else v))
          for k, v in P.items()}
    t0 = time.time()
    x = O.image_transform(im, **O.ROSS)
    feat = O.vgg_trunk(x, Pn["conv_w"], Pn["conv_b"])
    t_trunk = time.time() - t0
    t0 = time.time()
    b = boxes[:rois_sample]
    logits, deltas = O.frcnn_head(feat, O.project_im_rois(b, 1.0), Pn, chunk=500)
    scores = O.softmax(logits)
    dec = O.clamp_boxes(O.bbox_decode(b, deltas), W, H)
    t_head = (time.time() - t0) * (N_ROIS / float(rois_sample))
    t0 = time.time()
    for j in range(1, scores.shape[1]):
        sb, _ = O.select_scored(scores, dec, j, -1.5)
        (O.ref_nms if O.have_ref() else O.nms)(sb, 0.3)
    t_nms = (time.time() - t0) * (N_ROIS / float(rois_sample))
    total = t_trunk + t_head + t_nms
    return {"value": N_ROIS / total, "unit": "proposals/s", "cores": os.cpu_count(),
            "kind": "port", "sample": "1 image 600x1000: full trunk (%.1fs) + ROI head on %d of %d ROIs scaled (%.1fs) + NMS %s (%.2fs)"
            % (t_trunk, rois_sample, N_ROIS, t_head, "reference nms.c" if O.have_ref() else "port", t_nms),
            "seconds_per_image": total}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rois", type=int, default=250)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import multipathnet_amd
    from multipathnet_amd import models, parallel

    multipathnet_amd.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=N_CLASSES, seed=557)
    net = models.FastRCNN(P, max_h=H, max_w=W, max_rois=N_ROIS)
    im_np, boxes_np = synthetic_inputs()
    im, boxes = torch.from_numpy(im_np).to(dev), torch.from_numpy(boxes_np).to(dev)
    top_cap = net._dets.size(0)
    gathered = torch.empty((world, top_cap * 6 + 1), dtype=torch.float32, device=dev)

    pending = [None]

    def gather(bufs):
        if world > 1 and bufs is not None:  # RCCL gather of the scored-box record only (a few KB per rank)
            parallel.gather_detections(parallel.pack_record(bufs[0], bufs[1], top_cap), out=gathered)

    def step():
        # Tester:test loop form: image i's NMS/top-k tail runs on the pipeline's side stream and overlaps image
        # i+1's trunk; its detections are stream-ordered one call later, when they are gathered.
        cur = net.test_one_pipelined(im, boxes)
        gather(pending[0])
        pending[0] = cur

    def drain():
        net.flush()
        gather(pending[0])
        pending[0] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every step's tail and gather completes inside the timed region
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = args.steps * N_ROIS * world / dt

    # ---- roofline leg: same number of steps with HIP events around every kernel group
    net.set_profiling(True)
    net.get_profile(reset=True)
    for _ in range(args.steps):
        net.test_one_async(im, boxes)
    torch.cuda.synchronize()
    prof = net.get_profile(reset=True)
    net.set_profiling(False)

    if rank == 0:
        cf = conv_flops(models.VGG16_CFG, H, W)
        flops = {"conv_wino": sum(f for f, v in cf if v == "conv_wino"), "conv_direct": sum(f for f, v in cf if v == "conv_direct"),
                 "fc6": 2.0 * N_ROIS * 25088 * 4096, "fc7": 2.0 * N_ROIS * 4096 * 4096, "heads": 2.0 * N_ROIS * 4096 * 5 * N_CLASSES}
        launches = {"conv_wino": sum(1 for f, v in cf if v == "conv_wino"), "conv_direct": sum(1 for f, v in cf if v == "conv_direct")}
        kernels = {}
        for tag, (ms, cnt) in prof.items():
            if cnt:
                per_image_ms = ms / args.steps
                k = {"ms_per_image": round(per_image_ms, 4), "launches_per_image": cnt / args.steps}
                if tag in flops:
                    k["tflops"] = round(flops[tag] / (per_image_ms * 1e-3) / 1e12, 2)
                    k["frac_of_fp32_mfma_peak"] = round(flops[tag] / (per_image_ms * 1e-3) / FP32_MFMA_PEAK, 4)
                kernels[tag] = k
        dom = max((t for t in kernels if t in flops), key=lambda t: kernels[t]["ms_per_image"])
        n_launch = launches.get(dom, 1)
        achieved = flops[dom] / (kernels[dom]["ms_per_image"] * 1e-3) / 1e12
        total_flops = sum(f for f, _ in cf) + N_ROIS * 2.0 * (25088 * 4096 + 4096 * 4096 + 4096 * 105)
        out = {
            "metric": "proposals/sec (1000 ROIs, 600x1000 img) VGG-16 Fast R-CNN",
            "value": round(value, 1), "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VGG-16 Fast R-CNN, 1 image 600x1000 x 1000 ROIs per GPU per step, 21 classes, NMS 0.3, top-100 (BASELINE configs[1])",
                       "parallelism": "image-sharded x%d, all-gather of scored boxes only" % world},
            "whole_path_frac_of_fp32_mfma_peak": round(value / world * (total_flops / N_ROIS) / FP32_MFMA_PEAK, 4),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(achieved * 1e12 / FP32_MFMA_PEAK, 4), "traffic": None,
                         "flops_per_launch": flops[dom] / n_launch,
                         "avg_launch_ms": round(kernels[dom]["ms_per_image"] / n_launch, 5),
                         "how": "HIP events on the launch stream around each kernel group, %d profiled steps after the timed region" % args.steps},
            "kernels": kernels,
        }
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC passes cannot run inside the timed process: measured offline
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if dom in tj:
                out["roofline"]["traffic"] = tj[dom]["bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)"
                out["roofline"]["traffic_source"] = tj.get("_source")
        if dom == "conv_wino":  # the MFMA pipe executes 16 multiplies per 4 outputs instead of 36
            out["roofline"]["algorithm"] = ("Winograd F(2x2,3x3) in fp32: 'achieved' counts the ALGORITHMIC flops (2*H*W*Cin*9*Cout); the matrix "
                                            "pipe executes 1/2.25 of them, see executed_*")
            out["roofline"]["executed_tflops"] = round(achieved / 2.25, 2)
            out["roofline"]["executed_frac_of_peak"] = round(achieved / 2.25 * 1e12 / FP32_MFMA_PEAK, 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, im_np, boxes_np, args.cpu_rois)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
