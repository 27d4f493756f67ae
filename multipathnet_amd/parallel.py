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

    def record_floats(self, top_cap):
        return int(self._lib.mpn_det_record_floats(int(top_cap)))

    def gather_dets(self, dets, n_dets, out=None):
        """dets [top_cap,6], n_dets int32[1] (a test_one output pair) -> out [world, top_cap*6+1]; stream-ordered."""
        top_cap = dets.size(0)
        if out is None:
            out = torch.empty((self.world, self.record_floats(top_cap)), dtype=torch.float32, device=dets.device)
        _lib.check(self._lib.mpn_gather_dets(self._h, _f(dets), _i(n_dets), top_cap, _f(out), _stream()), "mpn_gather_dets")
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
    """[top_cap*6 + 1] fp32 record: rows beyond n are zero, last element = count."""
    rec = torch.zeros(top_cap * 6 + 1, dtype=torch.float32, device=dets.device)
    n = (n_dets.to(dets.device).reshape(()) if isinstance(n_dets, torch.Tensor) else torch.tensor(n_dets, device=dets.device)).clamp(0, top_cap)
    live = (torch.arange(top_cap, device=dets.device) < n).to(torch.float32).unsqueeze(1)  # no host sync: n stays on the device
    rec[: top_cap * 6] = (dets[:top_cap] * live).reshape(-1)
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
