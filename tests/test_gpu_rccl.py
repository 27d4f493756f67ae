"""The scored-box gather of the C ABI (mpn_comm_*, mpn_gather_dets: pack kernel + ncclAllGather bound from librccl at run
time) — what replaces test_runner.lua:91-104's result hand-back.

  * one GPU: a real one-rank RCCL communicator (the collective code path, stream-ordered) and the RCCL-free world-1 form;
  * two GPUs (skipped on a one-GPU box): two PROCESSES with a file-exchanged unique id (mpn_comm_init_rank), and the
    reference's own process model — two worker THREADS in one process (mpn_comm_init_all, test_runner.lua:55-66).
The sharding / merge logic around it is covered on CPU by tests/test_dist_gloo.py."""
import ctypes as C
import os
import tempfile
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dets(dev, top_cap, n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.rand((top_cap, 6), generator=g).to(dev)
    return d, torch.tensor([n], dtype=torch.int32, device=dev)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("use_rccl", [False, True])
def test_gather_dets_world1(dev, use_rccl):
    from multipathnet_amd import parallel
    comm = parallel.Comm.single(use_rccl=use_rccl)
    try:
        top_cap = 464
        for n in (100, 0, 464, 9999):  # 9999: a count beyond the capacity is clipped in the record
            dets, nd = _dets(dev, top_cap, n, n)
            out = comm.gather_dets(dets, nd)
            torch.cuda.synchronize()
            assert out.shape == (1, top_cap * 6 + 1)
            k = min(n, top_cap)
            assert int(out[0, -1].item()) == k
            rows = out[0, : top_cap * 6].view(top_cap, 6)
            assert torch.equal(rows[:k], dets[:k]) and (k == top_cap or float(rows[k:].abs().max()) == 0.0)
            assert torch.equal(parallel.unpack_record(out[0], top_cap), dets[:k])
            assert torch.equal(out[0], parallel.pack_record(dets, torch.tensor([k]), top_cap))  # same record as the host-side packer
    finally:
        comm.close()


def _proc_worker(rank, world, id_path, top_cap, q):
    import time
    import multipathnet_amd
    from multipathnet_amd import _lib
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    lib = multipathnet_amd.load()
    idbuf = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.mpn_comm_get_unique_id(idbuf), "unique id")
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(idbuf))
        os.replace(id_path + ".tmp", id_path)  # out-of-band hand-over, as a Lua host would do it
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id")
            time.sleep(0.05)
        idbuf = (C.c_char * 128).from_buffer_copy(open(id_path, "rb").read())
    h = C.c_void_p()
    _lib.check(lib.mpn_comm_init_rank(idbuf, world, rank, C.byref(h)), "init_rank")
    rec = int(lib.mpn_det_record_floats(top_cap))
    ok = True
    for step in range(3):
        dets, nd = _dets(dev, top_cap, 10 * rank + step + 1, 100 * step + rank)
        out = torch.empty((world, rec), device=dev)
        _lib.check(lib.mpn_gather_dets(h, C.cast(dets.data_ptr(), _lib.f32p), C.cast(nd.data_ptr(), _lib.i32p), top_cap,
                                       C.cast(out.data_ptr(), _lib.f32p), None), "gather")
        torch.cuda.synchronize()
        for r in range(world):
            er, _ = _dets(dev, top_cap, 0, 100 * step + r)
            k = 10 * r + step + 1
            ok = ok and int(out[r, -1].item()) == k and torch.equal(out[r, : k * 6].view(k, 6), er[:k])
    lib.mpn_comm_destroy(h)
    q.put((rank, ok))


@pytest.mark.timeout(600)
def test_gather_dets_two_processes(dev):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        id_path = os.path.join(d, "rccl_id")
        procs = [ctx.Process(target=_proc_worker, args=(r, 2, id_path, 64, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()  # the exact child we started
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.timeout(600)
def test_gather_dets_two_threads_one_process(dev):
    """test_runner.lua:55-66: Threads(nGPU), cutorch.setDevice(i) per thread, ONE process."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import multipathnet_amd
    from multipathnet_amd import _lib
    lib = multipathnet_amd.load()
    world, top_cap = 2, 64
    comms = (C.c_void_p * world)()
    devs = (C.c_int * world)(0, 1)
    _lib.check(lib.mpn_comm_init_all(world, devs, comms), "init_all")
    rec = int(lib.mpn_det_record_floats(top_cap))
    results = [None] * world

    def worker(r):
        torch.cuda.set_device(r)
        d = torch.device("cuda", r)
        dets, nd = _dets(d, top_cap, 5 + r, r)
        out = torch.empty((world, rec), device=d)
        rc = lib.mpn_gather_dets(C.c_void_p(comms[r]), C.cast(dets.data_ptr(), _lib.f32p), C.cast(nd.data_ptr(), _lib.i32p), top_cap,
                                 C.cast(out.data_ptr(), _lib.f32p), None)
        torch.cuda.synchronize(d)
        ok = rc == 0
        for q in range(world):
            eq, _ = _dets(d, top_cap, 0, q)
            ok = ok and int(out[q, -1].item()) == 5 + q and torch.equal(out[q, : (5 + q) * 6].view(5 + q, 6), eq[: 5 + q])
        results[r] = ok

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for r in range(world):
        lib.mpn_comm_destroy(C.c_void_p(comms[r]))
    assert results == [True, True]


def _run_bench(extra, env_extra, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]   # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_share_mode(dev):
    """`bench.py --gpus 2` end to end on a one-GPU box — self-launch under torch.distributed.run, RANK / LOCAL_RANK /
    WORLD_SIZE parsing, the gloo bootstrap group, the drift stream, the MAX-reduce of the elapsed time, rank-0-only printing.
    MPN_BENCH_SHARE_GPU=1 maps both ranks to device 0 and routes the record gather through gloo (RCCL refuses two ranks on one
    device); the line says so and is not a multi-GPU measurement."""
    out = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], {"MPN_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert "MPN_BENCH_SHARE_GPU=1" in out["config"]["parallelism"] and "NOT a multi-GPU measurement" in out["config"]["parallelism"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 1000 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-3 * out["value"]
    assert out["roofline"]["frac"] <= 1.0
    assert out["ranks"]["rccl_ranks"] == 0 and len(out["ranks"]["per_rank_proposals_per_s"]["all"]) == 2
    assert out["sustained"]["steps"] >= 800 and out["sustained"]["seconds"] >= 2.0   # the steady-state leg runs under two ranks too


@pytest.mark.timeout(900)
def test_bench_two_ranks_fallback_group_branch(dev):
    """bench.py's fallback — a torch.distributed group for the record gather when the C-ABI communicator does not come
    up — had never executed.  MPN_BENCH_FORCE_COMM_FAIL=1 sends every rank down that branch (the ranks agree on it through the MIN
    all-reduce, the group is created, the gather runs through torch.distributed); on a one-GPU box the group is gloo (RCCL refuses two
    ranks on one device), on a multi-GPU node the same branch creates the RCCL group.  The line says which ran, carries the per-rank
    rates and rccl_ranks = 0."""
    out = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--sustained-seconds", "0"],
                     {"MPN_BENCH_SHARE_GPU": "1", "MPN_BENCH_FORCE_COMM_FAIL": "1"})
    par = out["config"]["parallelism"]
    assert "fallback group" in par and "MPN_BENCH_FORCE_COMM_FAIL=1" in par and "NOT a multi-GPU measurement" in par
    assert out["ranks"]["rccl_ranks"] == 0 and len(out["ranks"]["per_rank_proposals_per_s"]["all"]) == 2
    assert out["ranks"]["per_rank_proposals_per_s"]["min"] * 2 <= out["value"] * 1.001  # the job's rate is bounded by its slowest rank
    assert out["sustained"] is None


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("mode", ["throughput", "latency"])
def test_bench_eight_ranks_share_mode(dev, mode):
    """VERDICT r4 task 8: `bench.py --gpus 8` — the world size the driver's scaling run uses (test_runner.lua:55-66: one worker per GPU) — had
    never been launched in any form.  Share mode (all eight ranks on device 0, the record gather through gloo; the line says so and is NOT a
    multi-GPU measurement) shakes out the self-launch, the NUMA / env parsing, the drift stream and the reductions at that world size, in the
    image-sharded throughput mode and in the proposal-sharded latency mode (every rank ends with identical detections)."""
    extra = ["--gpus", "8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--sustained-seconds", "0"]
    if mode == "latency":
        extra += ["--mode", "latency"]
    out = _run_bench(extra, {"MPN_BENCH_SHARE_GPU": "1"}, timeout=1400)
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["value"] > 0
    par = out["config"]["parallelism"]
    assert "MPN_BENCH_SHARE_GPU=1" in par and "NOT a multi-GPU measurement" in par
    assert out["ranks"]["rccl_ranks"] == 0
    if mode == "throughput":
        assert out["scaling"] == "weak" and len(out["ranks"]["per_rank_proposals_per_s"]["all"]) == 8
        assert out["ranks"]["per_rank_proposals_per_s"]["min"] * 8 <= out["value"] * 1.001
    else:
        assert out["ranks"]["final_detections_identical_on_all_ranks"] is True


def test_comm_reports_what_rccl_built(dev):
    """mpn_comm_rccl_ranks: ncclCommCount of the communicator (cross-checked against the caller's world / rank at init); 0 without RCCL"""
    from multipathnet_amd import parallel
    c0, c1 = parallel.Comm.single(use_rccl=False), parallel.Comm.single(use_rccl=True)
    try:
        assert c0.rccl_ranks == 0 and c1.rccl_ranks == 1
    finally:
        c0.close(); c1.close()


@pytest.mark.timeout(300)
def test_comm_init_is_bounded_in_time(dev):
    """a peer that never arrives must not hang the caller (and the GPU lease): ncclCommInitRank runs on a helper thread and
    mpn_comm_init_rank gives up after MPN_COMM_INIT_TIMEOUT_S with MPN_ENCCL and a message.  Run in a child process: the abandoned
    helper thread stays blocked inside RCCL until the process exits."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes as C, sys, time; sys.path.insert(0, %r)\n"
            "import torch, multipathnet_amd\n"
            "torch.cuda.set_device(0); lib = multipathnet_amd.load()\n"
            "idb = (C.c_char * 128)(); assert lib.mpn_comm_get_unique_id(idb) == 0\n"
            "h = C.c_void_p(); t0 = time.time()\n"
            "rc = lib.mpn_comm_init_rank(idb, 2, 0, C.byref(h))   # rank 1 never comes\n"
            "lib.mpn_last_error.restype = C.c_char_p\n"
            "print('RC', rc, round(time.time() - t0, 1), lib.mpn_last_error().decode()); sys.stdout.flush()\n"
            "import os; os._exit(0)\n") % root
    env = dict(os.environ, MPN_COMM_INIT_TIMEOUT_S="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    out = r.stdout.decode()
    line = [ln for ln in out.splitlines() if ln.startswith("RC ")]
    assert line, (out[-1000:], r.stderr.decode()[-1000:])
    _, rc, secs, msg = line[0].split(" ", 3)
    assert int(rc) != 0 and 3.0 <= float(secs) <= 60.0 and "did not return within 4 s" in msg
    assert time.time() - t0 < 200


@pytest.mark.timeout(900)
def test_bench_latency_mode_single_rank(dev):
    """`bench.py --mode latency` at N = 1: mpn_frcnn_test_one_sharded over a world-1 communicator, its own metric string, plus the
    one-GPU projection of rank 0's share of an 8-rank world."""
    out = _run_bench(["--mode", "latency", "--steps", "5", "--warmup", "2"], {})
    assert "latency" in out["metric"] and "not the headline" in out["metric"] and out["higher_is_better"] is False and out["scaling"] == "strong"
    assert out["unit"] == "ms/image" and 0 < out["value"] < 50 and out["unsharded_ms"] > 0
    assert out["projected"]["world"] == 8 and 0 < out["projected"]["rank0_compute_ms"] < out["value"]
    assert out["ranks"]["final_detections_identical_on_all_ranks"] is True and out["ranks"]["n_detections"] > 0
