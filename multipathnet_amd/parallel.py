"""Multi-GPU layer: image-sharded replicas + a gather of SCORED BOXES ONLY.

Replaces the reference's two mechanisms: `test_runner.lua:55-66,91-104` (one thread +
full model replica per GPU, one job per image, results serialised back to the main thread) and
`ModelParallelTable.lua:195-242` (broadcast of whole feature maps to tower GPUs).  Here: one process
per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests), rank r
owns images r, r+G, r+2G, ...; weights are resident per rank; the only traffic is one fixed-size
record per image — `top_cap` rows of {x1,y1,x2,y2,score,class} plus a count — all-gathered, i.e. a
few KB, latency-bound, never features or logits.
"""
import torch
import torch.distributed as dist


def shard_indices(n_images, rank, world):
    """test_runner.lua:91-104 partitioning contract: image i goes to worker i % world."""
    return list(range(rank, n_images, world))


def pack_record(dets, n_dets, top_cap):
    """[top_cap*6 + 1] fp32 record: rows beyond n are zero, last element = count."""
    rec = torch.zeros(top_cap * 6 + 1, dtype=torch.float32, device=dets.device)
    rec[: top_cap * 6] = dets[:top_cap].reshape(-1)
    rec[-1] = n_dets.to(torch.float32).reshape(()) if isinstance(n_dets, torch.Tensor) else float(n_dets)
    return rec


def unpack_record(rec, top_cap):
    n = int(rec[-1].item())
    return rec[: top_cap * 6].view(top_cap, 6)[: min(n, top_cap)]


def gather_detections(rec, group=None, out=None):
    """All-gather one record per rank (stream-ordered on RCCL; no host sync).  Returns [world, len(rec)]."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, rec.numel()), dtype=rec.dtype, device=rec.device)
    if world == 1:
        out[0].copy_(rec)
        return out
    dist.all_gather_into_tensor(out.view(-1), rec, group=group)
    return out


def merge_by_image(gathered_per_step, world, n_images, top_cap):
    """Re-interleaves per-step gathers back into image order: step t, rank r -> image t*world + r."""
    out = [None] * n_images
    for t, g in enumerate(gathered_per_step):
        for r in range(world):
            i = t * world + r
            if i < n_images:
                out[i] = unpack_record(g[r], top_cap)
    return out
