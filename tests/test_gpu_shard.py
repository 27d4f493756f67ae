"""Proposal (ROI) sharding of ONE image — the latency mode (mpn_frcnn_shard_* / mpn_frcnn_test_one_sharded, include/mpn.h;
replaces ModelParallelTable.lua:195-242 for a single image).  A one-GPU box cannot run two RCCL ranks, so the G-rank exchange is
emulated on ONE device: rank r = 0..G-1's steps run one after the other on one handle and the all-gather is the concatenation of
their records — everything except the ncclAllGather call itself (covered with one real RCCL rank below and with two ranks where
two GPUs exist).  The bar: the sharded result equals the unsharded mpn_frcnn_test_one's BIT FOR BIT — rows are independent
(row-invariant GEMM summation, dense.h linear_c8) and classes are independent."""
import numpy as np
import pytest
import torch

from test_gpu_pipeline import SMALL, _boxes

pytestmark = pytest.mark.gpu


def _emulate(net, im, boxes, world):
    """all G ranks' steps on one device; returns (dets, n, rows_all, class_all)"""
    N = boxes.size(0)
    rr, cr = net.shard_record_floats(N, world)
    rows_all = torch.empty((world, rr), dtype=torch.float32, device=im.device)
    class_all = torch.empty((world, cr), dtype=torch.float32, device=im.device)
    for r in range(world):
        net.shard_head(im, boxes, r, world, out=rows_all[r])
    for r in range(world):
        net.shard_nms(rows_all, N, r, world, out=class_all[r])
    dets, n = net.shard_finish(class_all, N, world)
    torch.cuda.synchronize()
    return dets[: int(n.item())].clone(), int(n.item()), rows_all, class_all


def _reference(net, im, boxes):
    dets, n = net.test_one_async(im, boxes)
    torch.cuda.synchronize()
    keep, kidx, nk = [t.clone() for t in net.nms_results()]
    return dets[: int(n.item())].clone(), keep, kidx, nk


@pytest.fixture(scope="module")
def small_nets(dev):
    from multipathnet_amd import models
    s = SMALL
    P = models.synthetic_params(s["cfg"], pooled=7, fc_dim=s["fc"], n_classes=s["C"], seed=557)
    mk = lambda **kw: models.FastRCNN(P, cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"], **kw)
    rng = np.random.default_rng(555)
    im = torch.from_numpy(rng.random((3, s["H"], s["W"]), dtype=np.float32)).to(dev)
    return dict(plain=mk(), iter2vote=mk(num_iter=2, bbox_voting=True, bbox_vote_thresh=0.5, bbox_vote_score_pow=0.5),
                iter3rbox=mk(num_iter=3, use_rbox_scores=True), im=im)


@pytest.mark.parametrize("kind", ["plain", "iter2vote", "iter3rbox"])
@pytest.mark.parametrize("world,N", [(1, 200), (2, 200), (3, 200), (8, 200), (4, 37), (8, 5), (7, 6)])
def test_sharded_equals_unsharded_emulated(dev, small_nets, kind, world, N):
    """ragged N (200 over 3, 37 over 4), more ranks than proposals (5 over 8: three ranks own nothing), 6 foreground classes over
    4 / 7 / 8 ranks (ranks without a class), iterative localisation (each rank refines its own rows; pass-major joined tables),
    box voting with a score exponent, test_use_rbox_scores."""
    from multipathnet_amd import parallel
    s = SMALL
    net, im = small_nets[kind], small_nets["im"]
    boxes = torch.from_numpy(_boxes(np.random.default_rng(1000 + N), N, s["W"], s["H"])).to(dev)
    ref_dets, keep, kidx, nk = _reference(net, im, boxes)
    if kind == "plain":
        sc_ref, bb_ref = net.detect(im, boxes)
    dets, n, rows_all, class_all = _emulate(net, im, boxes, world)
    keep2, kidx2, nk2 = net.nms_results()
    assert torch.equal(nk2, nk)
    C1 = s["C"] - 1
    for c in range(C1):
        k = int(nk[c])
        assert torch.equal(keep2[c, :k], keep[c, :k]) and torch.equal(kidx2[c, :k], kidx[c, :k]), c
    assert dets.shape == ref_dets.shape and torch.equal(dets, ref_dets)
    # the records are the layouts parallel.py documents (and tests/test_dist_gloo.py exchanges under gloo)
    P = {"plain": 1, "iter2vote": 2, "iter3rbox": 2}[kind]
    sc, bb = parallel.unpack_rows_records(rows_all, N, world, P, s["C"])
    if kind == "plain":
        assert torch.equal(sc, sc_ref) and torch.equal(bb, bb_ref)   # the whole image's tables, row for row
    K, KI, NK, V = parallel.unpack_class_records(class_all, C1, world, P * N, voting=(kind == "iter2vote"))
    assert torch.equal(NK, nk)
    for c in range(C1):
        k = int(nk[c])
        assert torch.equal(KI[c, :k], kidx[c, :k])
        assert torch.equal((V if kind == "iter2vote" else K)[c, :k], keep[c, :k])   # nms_results returns the voted table when voting


def test_rows_are_invariant_to_the_batch_they_are_scored_in_fullsize(dev):
    """memoryEfficientForward's property (ImageDetect.lua:126-133: chunked == un-chunked EXACTLY) at BASELINE size, across the
    launch forms of the head GEMMs: 1000 rows run fc6 / fc7 un-split, 125 rows (an 8-GPU shard) run them split-K — both follow
    the canonical segment order, so a row's scores do not depend on the rows it is batched with."""
    import bench
    from multipathnet_amd import models
    P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557, head_scale="trained")
    net = models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    im, boxes = bench.synthetic_inputs()
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    s, b = net.detect(imd, bd)
    for lo, hi in [(0, 125), (375, 500), (0, 250), (500, 1000), (999, 1000), (3, 390)]:
        s2, b2 = net.detect(imd, bd[lo:hi].contiguous(), recompute_features=False)
        assert torch.equal(s2, s[lo:hi]) and torch.equal(b2, b[lo:hi]), (lo, hi)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_equals_unsharded_emulated_fullsize(dev, world):
    """BASELINE configs[1] (600x1000, 1000 ROIs, 21 classes) sharded over 2 and 8 emulated ranks == one GPU, bit for bit,
    at the trained score scale (saturating softmax rows)."""
    import bench
    from multipathnet_amd import models
    P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557, head_scale="trained")
    net = models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    im, boxes = bench.synthetic_inputs()
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    ref_dets, keep, kidx, nk = _reference(net, imd, bd)
    dets, n, _, _ = _emulate(net, imd, bd, world)
    keep2, kidx2, nk2 = net.nms_results()
    assert torch.equal(nk2, nk) and n == ref_dets.shape[0] and torch.equal(dets, ref_dets)
    for c in range(bench.N_CLASSES - 1):
        k = int(nk[c])
        assert torch.equal(keep2[c, :k], keep[c, :k]) and torch.equal(kidx2[c, :k], kidx[c, :k])


def test_one_sharded_over_a_real_one_rank_rccl_communicator(dev, small_nets):
    """mpn_frcnn_test_one_sharded end to end (both exchanges through ncclAllGather on a world-1 RCCL communicator, and through
    the RCCL-free world-1 communicator) == mpn_frcnn_test_one."""
    from multipathnet_amd import parallel
    s = SMALL
    im = small_nets["im"]
    boxes = torch.from_numpy(_boxes(np.random.default_rng(5), 150, s["W"], s["H"])).to(dev)
    for kind in ("plain", "iter2vote"):
        net = small_nets[kind]
        ref_dets, keep, kidx, nk = _reference(net, im, boxes)
        for use_rccl in (False, True):
            comm = parallel.Comm.single(use_rccl=use_rccl)
            for _ in range(2):  # steady state: the second call allocates nothing
                dets, n = net.test_one_sharded(comm, im, boxes)
            torch.cuda.synchronize()
            assert torch.equal(dets[: int(n.item())], ref_dets)
            keep2, kidx2, nk2 = net.nms_results()
            assert torch.equal(nk2, nk)
            comm.close()


def _two_rank_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multipathnet_amd import models, parallel
    s = SMALL
    P = models.synthetic_params(s["cfg"], pooled=7, fc_dim=s["fc"], n_classes=s["C"], seed=557)
    net = models.FastRCNN(P, cfg=s["cfg"], pooled=7, spatial_scale=s["scale"], max_h=s["H"], max_w=s["W"], max_rois=s["N"])
    dev = torch.device("cuda", rank)
    im = torch.from_numpy(np.random.default_rng(555).random((3, s["H"], s["W"]), dtype=np.float32)).to(dev)
    boxes = torch.from_numpy(_boxes(np.random.default_rng(9), 199, s["W"], s["H"])).to(dev)
    ref, n = net.test_one_async(im, boxes)
    torch.cuda.synchronize()
    ref = ref[: int(n.item())].clone()
    comm = parallel.Comm.from_torch_distributed()
    dets, n = net.test_one_sharded(comm, im, boxes)
    torch.cuda.synchronize()
    q.put((rank, bool(torch.equal(dets[: int(n.item())], ref))))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_one_sharded_two_gpus(dev):
    """two real ranks over RCCL (skips on a one-GPU box)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from test_dist_gloo import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert sorted(got) == [(0, True), (1, True)]


@pytest.mark.parametrize("world", [2, 5, 8])
def test_multipathnet_sharded_equals_unsharded_emulated(dev, world):
    """The mode pays off most where the per-ROI head dominates: MultiPathNet (BASELINE configs[2]; 5 towers, skip pooling, integral
    heads).  Same bar: bit-identical to the unsharded test_one — the mix / fc6 / fc7 / integral-head GEMMs are row-invariant, the
    skip pooling and the L2 normalisation are per ROI."""
    from multipathnet_amd import models
    cfg = [8, 16, "P", 16, 24, "P", 32, 32, "P", 64, "P", 64]
    H, W, N, Cn, K = 150, 250, 117, 9, 3
    P = models.synthetic_mpnet_params(cfg, pooled=7, fc_dim=128, n_classes=Cn, n_integral=K, seed=11)
    rng = np.random.default_rng(21)
    im = torch.from_numpy(rng.random((3, H, W), dtype=np.float32)).to(dev)
    boxes = torch.from_numpy(_boxes(rng, N, W, H, lo=12)).to(dev)
    net = models.MultiPathNet(P, cfg=cfg, pooled=7, spatial_scale=1 / 16, max_h=H, max_w=W, max_rois=N)
    sc_ref, bb_ref = net.detect(im, boxes)
    ref_dets, keep, kidx, nk = _reference(net, im, boxes)
    dets, n, rows_all, _ = _emulate(net, im, boxes, world)
    from multipathnet_amd import parallel
    sc, bb = parallel.unpack_rows_records(rows_all, N, world, 1, Cn)
    assert torch.equal(sc, sc_ref) and torch.equal(bb, bb_ref)
    keep2, kidx2, nk2 = net.nms_results()
    assert torch.equal(nk2, nk) and torch.equal(dets, ref_dets)
    for c in range(Cn - 1):
        k = int(nk[c])
        assert torch.equal(keep2[c, :k], keep[c, :k]) and torch.equal(kidx2[c, :k], kidx[c, :k])


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("model", ["resnet", "resnet_towers", "inception"])
def test_graph_models_sharded_equal_unsharded_emulated(dev, model, bf16):
    """The graph pipelines (ResNet / Inception, plain and tower forms, fp32 and bf16): since round 4 everything that fixes the summation
    ORDER of a per-ROI layer is a function of the layer alone, never of the ROI count (resnet.hip rn_conv `per_roi`: no split-K, pointwise
    convolutions always on the un-split GEMM, 3x3 convolutions always on the Winograd mosaic with a batch-independent plan; tile shapes
    may still follow the batch), so a ROI's rows are bit-invariant to the shard it is scored in and the sharded mode equals the unsharded
    test_one bit for bit here too (ImageDetect.lua:126-133's chunked == full for ANY model; worlds 2 / 5 / 8, ragged N)."""
    from multipathnet_amd import models, parallel
    from test_gpu_resnet import _inputs
    if model == "inception":
        H, W, N, C = 150, 200, 53, 5
        G = models.synthetic_inception_v3_params(n_classes=C, width=0.125, seed=9)
        net = models.InceptionFRCNN(G, max_h=H, max_w=W, max_rois=64, top_k=10, bf16=bf16)
    else:
        H, W, N, C = 120, 160, 50, 5
        if model == "resnet":
            R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=8, blocks=[1, 1, 1, 2], block_type="bottleneck", seed=3)
        else:
            R = models.synthetic_resnet_mpn_params(depth=0, n_classes=C, n_integral=2, base_width=8, blocks=[1, 1, 1, 2], block_type="basic", seed=3)
        net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=10, bf16=bf16)
    im, boxes = _inputs(H, W, N, 9)
    imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
    sc_ref, bb_ref = net.detect(imd, bd)
    ref_dets, keep, kidx, nk = _reference(net, imd, bd)
    for world in (2, 5, 8):
        dets, n, rows_all, _ = _emulate(net, imd, bd, world)
        sc, bb = parallel.unpack_rows_records(rows_all, N, world, 1, C)
        assert torch.equal(sc, sc_ref) and torch.equal(bb, bb_ref), world
        keep2, kidx2, nk2 = net.nms_results()
        assert torch.equal(nk2, nk) and torch.equal(dets, ref_dets)
        for c in range(C - 1):
            k = int(nk[c])
            assert torch.equal(keep2[c, :k], keep[c, :k]) and torch.equal(kidx2[c, :k], kidx[c, :k])


def test_graph_models_round3_dispatch_was_not_batch_invariant(dev):
    """the hook that restores round 3's per-ROI dispatch (kernel / split-K by batch size) — the control for the test above: with it the
    same shards differ in the last bits, so the invariance is the dispatch rule's doing and the test can see a violation"""
    from conftest import hooks
    from multipathnet_amd import models
    from test_gpu_resnet import _inputs
    H, W, N, C = 120, 160, 50, 5
    R = models.synthetic_resnet_params(depth=0, n_classes=C, base_width=16, blocks=[1, 1, 1, 2], block_type="bottleneck", seed=3)
    im, boxes = _inputs(H, W, N, 9)
    with hooks(roi_invariant=0):
        net = models.ResNetFRCNN(R, max_h=H, max_w=W, max_rois=64, top_k=10)
        imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev)
        s, b = net.detect(imd, bd)
        s2, b2 = net.detect(imd, bd[:7].contiguous(), recompute_features=False)
        close = float((s2 - s[:7]).abs().max()) < 1e-5
        same = torch.equal(s2, s[:7]) and torch.equal(b2, b[:7])
        del net
    assert close and not same
