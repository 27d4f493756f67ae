"""RCCL smoke on one GPU: the exact collective bench.py issues for N>1 (all_gather_into_tensor of the scored-box record on
the device, stream-ordered) with backend "nccl" (== RCCL on ROCm) and world_size 1.  The real N>1 runs are the driver's;
the multi-rank logic is covered on CPU by tests/test_dist_gloo.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(180)
def test_rccl_all_gather_record_world1(dev):
    import torch.distributed as dist
    from multipathnet_amd import parallel
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        top_cap = 464
        dets = torch.rand((top_cap, 6), device=dev)
        n = torch.tensor([100], dtype=torch.int32, device=dev)
        rec = parallel.pack_record(dets, n, top_cap)
        out = torch.empty((1, rec.numel()), device=dev)
        dist.all_gather_into_tensor(out.view(-1), rec)   # what gather_detections does for world > 1
        torch.cuda.synchronize()
        assert torch.equal(out[0], rec)
        assert torch.equal(parallel.unpack_record(out[0], top_cap), dets[:100])
        dist.barrier()
    finally:
        dist.destroy_process_group()
