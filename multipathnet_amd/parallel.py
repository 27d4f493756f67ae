"""Multi-GPU layer: image-sharded replicas + a gather of SCORED BOXES ONLY.

Replaces the reference's two mechanisms: `test_runner.lua:55-66,91-104` (one thread +
full model replica per GPU, one job per image, results serialised back to the main thread) and
`ModelParallelTable.lua:195-242` (broadcast of whole feature maps to tower GPUs).  Here: one process
per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests), rank r
owns images r, r+G, r+2G, ...; weights are resident per rank; the only traffic is one fixed-size
record per image — `top_cap` rows of {x1,y1,x2,y2,score,class} plus a count — all-gathered, i.e. a
few KB, latency-bound, never features or logits.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .nn import _f, _i, _stream


class Comm(object):
    """One RCCL communicator rank behind the C ABI (mpn_comm_*, include/mpn.h) — what a Lua worker would hold.

    `Comm.from_torch_distributed()` uses torch.distributed only as the bootstrap: rank 0's 128-byte RCCL unique id is
    broadcast through the process group (any backend), then every rank calls mpn_comm_init_rank.  The data path
    (`gather_dets`) is mpn_gather_dets: pack kernel + ncclAllGather on the caller's stream, no torch collective."""

    def __init__(self, handle, world, rank, lib):
        self._h, self.world, self.rank, self._lib = handle, world, rank, lib

    @classmethod
    def single(cls, use_rccl=False):
        """world-size-1 communicator; use_rccl=True makes it a real one-rank RCCL communicator (tests)."""
        lib = _lib.load()
        idbuf = None
        if use_rccl:
            idbuf = (C.c_char * 128)()
            _lib.check(lib.mpn_comm_get_unique_id(idbuf), "mpn_comm_get_unique_id")
        h = C.c_void_p()
        _lib.check(lib.mpn_comm_init_rank(idbuf, 1, 0, C.byref(h)), "mpn_comm_init_rank")
        return cls(h, 1, 0, lib)

    @classmethod
    def from_torch_distributed(cls, group=None):
        lib = _lib.load()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        idbuf = (C.c_char * 128)()
        if world > 1:
            dev = torch.device("cuda", torch.cuda.current_device())
            if rank == 0:
                _lib.check(lib.mpn_comm_get_unique_id(idbuf), "mpn_comm_get_unique_id")
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.broadcast(t, src=0, group=group)
            idbuf = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
        h = C.c_void_p()
        _lib.check(lib.mpn_comm_init_rank(idbuf if world > 1 else None, world, rank, C.byref(h)), "mpn_comm_init_rank")
        return cls(h, world, rank, lib)

    @property
    def rccl_ranks(self):
        """the rank count RCCL itself reports for this communicator (ncclCommCount); 0 = no RCCL communicator (world 1 without an id)"""
        return int(self._lib.mpn_comm_rccl_ranks(self._h)) if self._h is not None and self._h.value else 0

    def record_floats(self, top_cap):
        return int(self._lib.mpn_det_record_floats(int(top_cap)))

    def gather_dets(self, dets, n_dets, out=None):
        """dets [top_cap,6], n_dets int32[1] (a test_one output pair) -> out [world, top_cap*6+1]; stream-ordered."""
        top_cap = dets.size(0)
        if out is None:
            out = torch.empty((self.world, self.record_floats(top_cap)), dtype=torch.float32, device=dets.device)
        _lib.check(self._lib.mpn_gather_dets(self._h, _f(dets), _i(n_dets), top_cap, _f(out), _stream()), "mpn_gather_dets")
        return out

    def gather_rows(self, send, out=None):
        """mpn_gather_rows: all-gather one fixed-size float record per rank -> [world, send.numel()]; stream-ordered."""
        n = send.numel()
        if out is None:
            out = torch.empty((self.world, n), dtype=torch.float32, device=send.device)
        _lib.check(self._lib.mpn_gather_rows(self._h, _f(send), n, _f(out), _stream()), "mpn_gather_rows")
        return out

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.mpn_comm_destroy(self._h)
        self._h = None

    __del__ = close


def shard_indices(n_images, rank, world):
    """test_runner.lua:91-104 partitioning contract: image i goes to worker i % world."""
    return list(range(rank, n_images, world))


def pack_record(dets, n_dets, top_cap):
    """[top_cap*6 + 1] fp32 record: rows beyond n are zero, last element = count (mirror of pack_det_record_kernel, comm.hip)."""
    rec = torch.zeros(top_cap * 6 + 1, dtype=torch.float32, device=dets.device)
    n = (n_dets.to(dets.device).reshape(()) if isinstance(n_dets, torch.Tensor) else torch.tensor(n_dets, device=dets.device)).clamp(0, top_cap)
    live = (torch.arange(top_cap, device=dets.device) < n).unsqueeze(1)  # no host sync: n stays on the device
    # SELECT zero for dead rows (as the kernel does): multiplying by 0 would let NaN / Inf garbage in them survive
    rec[: top_cap * 6] = torch.where(live, dets[:top_cap], torch.zeros((), dtype=dets.dtype, device=dets.device)).reshape(-1)
    rec[-1] = n.to(torch.float32)
    return rec


def unpack_record(rec, top_cap):
    n = int(rec[-1].item())
    return rec[: top_cap * 6].view(top_cap, 6)[: min(n, top_cap)]


def gather_detections(rec, group=None, out=None):
    """torch.distributed form of the gather (used by the gloo CPU tests of the sharding logic; the GPU path is Comm.gather_dets).
    All-gather one record per rank.  Returns [world, len(rec)]."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Proposal (ROI) sharding of ONE image (latency mode; mpn_frcnn_shard_* in include/mpn.h): host mirrors of the partition
# rule and of the two record layouts.  The device path is FastRCNN.test_one_sharded / shard_head / shard_nms /
# shard_finish; these torch forms exist so that the partition / merge logic runs under gloo on CPU
# (tests/test_dist_gloo.py) and so that the device's records can be checked element for element (tests/test_gpu_shard.py).
# ---------------------------------------------------------------------------------------------------------------------
def shard_range(n, world, rank):
    """mpn_shard_range: balanced contiguous [lo, hi) — the first n % world ranks own one item more (host-only C call)."""
    lo, hi = C.c_int(), C.c_int()
    _lib.check(_lib.load().mpn_shard_range(int(n), int(world), int(rank), C.byref(lo), C.byref(hi)), "mpn_shard_range")
    return lo.value, hi.value


def shard_chunk(n, world):
    return (n + world - 1) // world


def pack_rows_record(scores, bbox, n_local, n_passes, chunk):
    """scores [P*n_local, C], bbox [P*n_local, 4C] (pass-major joined tables of ONE rank) -> its row record
    [scores: P x chunk x C | boxes: P x chunk x 4C], rows at / beyond n_local zero (mirror of shard_pack_rows_kernel)."""
    Cc = scores.shape[1]
    rs = torch.zeros((n_passes, chunk, Cc), dtype=torch.float32, device=scores.device)
    rb = torch.zeros((n_passes, chunk, 4 * Cc), dtype=torch.float32, device=scores.device)
    if n_local:
        rs[:, :n_local] = scores.view(n_passes, n_local, Cc)
        rb[:, :n_local] = bbox.view(n_passes, n_local, 4 * Cc)
    return torch.cat([rs.reshape(-1), rb.reshape(-1)])


def unpack_rows_records(rows_all, N, world, n_passes, n_classes):
    """[world, rows_floats] -> the image's joined tables (scores [P*N, C], bbox [P*N, 4C]) in the unsharded row order
    (mirror of shard_unpack_rows_kernel)."""
    chunk, Cc = shard_chunk(N, world), n_classes
    sc = torch.empty((n_passes, N, Cc), dtype=torch.float32, device=rows_all.device)
    bb = torch.empty((n_passes, N, 4 * Cc), dtype=torch.float32, device=rows_all.device)
    ns = n_passes * chunk * Cc
    for r in range(world):
        lo, hi = shard_range(N, world, r)
        if hi > lo:
            sc[:, lo:hi] = rows_all[r, :ns].view(n_passes, chunk, Cc)[:, : hi - lo]
            bb[:, lo:hi] = rows_all[r, ns:].view(n_passes, chunk, 4 * Cc)[:, : hi - lo]
    return sc.view(n_passes * N, Cc), bb.view(n_passes * N, 4 * Cc)


def class_record_floats(n_fg_classes, world, rows, voting=False):
    cmax = shard_chunk(n_fg_classes, world)
    return cmax * (1 + rows * (11 if voting else 6))


def pack_class_record(keep, keep_idx, n_keep, c0, c1, rows, cmax, voted=None):
    """keep [n_cls, rows, 5], keep_idx [n_cls, rows] int32, n_keep [n_cls] int32 (only classes c0..c1-1 need be valid) ->
    [n_keep: cmax ints | keep: cmax x rows x 5 | keep_idx: cmax x rows ints | voted: cmax x rows x 5] with the integers stored
    bit for bit in float slots; rows at / beyond a class's n_keep zero (the device leaves them unwritten; nothing reads them)."""
    dev = keep.device
    hdr = torch.zeros(cmax, dtype=torch.int32, device=dev)
    k = torch.zeros((cmax, rows, 5), dtype=torch.float32, device=dev)
    ki = torch.zeros((cmax, rows), dtype=torch.int32, device=dev)
    v = torch.zeros((cmax, rows, 5), dtype=torch.float32, device=dev) if voted is not None else None
    for j, c in enumerate(range(c0, c1)):
        n = int(n_keep[c])
        hdr[j] = n
        k[j, :n] = keep[c, :n]
        ki[j, :n] = keep_idx[c, :n]
        if v is not None:
            v[j, :n] = voted[c, :n]
    parts = [hdr.view(torch.float32), k.reshape(-1), ki.view(torch.float32).reshape(-1)]
    if v is not None:
        parts.append(v.reshape(-1))
    return torch.cat(parts)


def unpack_class_records(class_all, n_fg_classes, world, rows, voting=False):
    """[world, class_floats] -> (keep [n_cls, rows, 5], keep_idx [n_cls, rows], n_keep [n_cls], voted or None); rows beyond
    n_keep are zero (mirror of shard_unpack_classes_kernel)."""
    cmax = shard_chunk(n_fg_classes, world)
    dev = class_all.device
    keep = torch.zeros((n_fg_classes, rows, 5), dtype=torch.float32, device=dev)
    kidx = torch.zeros((n_fg_classes, rows), dtype=torch.int32, device=dev)
    nk = torch.zeros(n_fg_classes, dtype=torch.int32, device=dev)
    voted = torch.zeros_like(keep) if voting else None
    for r in range(world):
        c0, c1 = shard_range(n_fg_classes, world, r)
        rec = class_all[r]
        hdr = rec[:cmax].view(torch.int32)
        k = rec[cmax: cmax + cmax * rows * 5].view(cmax, rows, 5)
        ki = rec[cmax + cmax * rows * 5: cmax + cmax * rows * 6].view(torch.int32).view(cmax, rows)
        v = rec[cmax + cmax * rows * 6:].view(cmax, rows, 5) if voting else None
        for j, c in enumerate(range(c0, c1)):
            n = int(hdr[j])
            nk[c] = n
            keep[c, :n] = k[j, :n]
            kidx[c, :n] = ki[j, :n]
            if voting:
                voted[c, :n] = v[j, :n]
    return keep, kidx, nk, voted
