"""BASELINE configs[1] at FULL size (VGG-16 Fast R-CNN, 600x1000 image, 1000 ROIs, 21 classes, the bench's own synthetic
inputs and weights): (1) parity against the oracle on a ROI sample — the oracle runs the whole trunk (a few seconds on the GPU
box's host cores) and the head for 100 of the 1000 ROIs — class logits and box deltas to 1e-4 absolute — plus an
oracle-independent PyTorch-CPU cross-check; (2) the size-independent properties of the path on all 1000 ROIs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(dev):
    import bench
    from multipathnet_amd import models
    P = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557)
    net = models.FastRCNN(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    im, boxes = bench.synthetic_inputs()
    return dict(P=P, net=net, im=im, boxes=boxes, imd=torch.from_numpy(im).to(dev), bd=torch.from_numpy(boxes).to(dev))


N_SAMPLE = 100  # ROIs whose head the oracle evaluates (rows are independent; < 2 s on the GPU box's host cores)


def _np_tree(P):
    return {k: ([t.numpy() for t in v] if isinstance(v, list) and v and hasattr(v[0], "numpy") else (v.numpy() if hasattr(v, "numpy") else v))
            for k, v in P.items()}


def test_fullsize_logits_deltas_scores_vs_oracle_on_roi_sample(O, dev, full):
    """north_star: 'class scores / bbox-regression within 1e-4 fp32'.  Checked at full size on the raw quantities — the class
    LOGITS and the bbox-regression DELTAS (after BBoxNorm), absolute 1e-4, not only on the softmax scores (a softmax
    compresses logit error by p(1-p)) — for a 100-ROI sample, plus scores and decoded boxes."""
    net, im, boxes, P = full["net"], full["im"], full["boxes"], full["P"]
    s, b = net.detect(full["imd"], full["bd"])
    s, b = s.cpu().numpy(), b.cpu().numpy()
    N, C = boxes.shape[0], net.n_classes
    cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
    Pn = _np_tree(P)
    x = O.image_transform(im, **O.ROSS)
    feat = O.vgg_trunk(x, Pn["conv_w"], Pn["conv_b"])  # full 600x1000 trunk on the host cores
    conv5 = net.debug_tensor("conv5", (512, feat.shape[1], feat.shape[2])).cpu().numpy()
    assert feat.shape == conv5.shape == (512, 38, 63)
    assert np.abs(conv5 - feat).max() < 1e-4 * max(1.0, np.abs(feat).max())
    idx = np.random.default_rng(7).choice(N, N_SAMPLE, replace=False)
    rois = O.project_im_rois(boxes[idx], 1.0)
    logits, deltas = O.frcnn_head(feat, rois, Pn)
    e_logit, e_delta = np.abs(cls[idx] - logits).max(), np.abs(raw[idx] - deltas).max()
    print("full-size head, %d ROIs: max|dlogit| = %.3g (|logit| <= %.3g), max|ddelta| = %.3g (|delta| <= %.3g)"
          % (N_SAMPLE, e_logit, np.abs(logits).max(), e_delta, np.abs(deltas).max()))
    assert e_logit < 1e-4 and e_delta < 1e-4          # ABSOLUTE, on the pre-softmax / pre-decode quantities
    so = O.softmax(logits)
    assert np.abs(s[idx] - so).max() < 1e-4           # class scores
    bo = O.clamp_boxes(O.bbox_decode(boxes[idx], deltas), im.shape[2], im.shape[1])
    assert np.abs(b[idx] - bo).max() < 5e-3           # decoded boxes, pixels


def test_fullsize_trunk_and_head_vs_pytorch_cpu(O, dev, full):
    """An oracle-independent cross-check of the dense arithmetic at full size: PyTorch-CPU (oneDNN) conv2d / max_pool2d(ceil_mode)
    / linear — the cudnn / nn semantics the reference relies on — on the bench's image and weights.  Only the ROI pooling of the
    sample (integer bin bounds + max, no arithmetic) comes from the oracle."""
    import torch.nn.functional as F
    from multipathnet_amd import models
    net, im, boxes, P = full["net"], full["im"], full["boxes"], full["P"]
    net.detect(full["imd"], full["bd"])
    N, C = boxes.shape[0], net.n_classes
    with torch.no_grad():
        x = torch.from_numpy(O.image_transform(im, **O.ROSS)).unsqueeze(0)
        li = 0
        for item in models.VGG16_CFG:
            if item == "P":
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            else:
                x = F.relu(F.conv2d(x, P["conv_w"][li], P["conv_b"][li], padding=1))
                li += 1
        feat = x[0].numpy()
        conv5 = net.debug_tensor("conv5", feat.shape).cpu().numpy()
        assert np.abs(conv5 - feat).max() < 1e-4 * max(1.0, np.abs(feat).max())
        idx = np.random.default_rng(11).choice(N, N_SAMPLE, replace=False)
        pooled, _ = O.roi_pool(feat, O.project_im_rois(boxes[idx], 1.0), 7, 7, 1.0 / 16)
        h = torch.from_numpy(pooled.reshape(N_SAMPLE, -1))
        h = F.relu(F.linear(h, P["fc6_w"], P["fc6_b"]))
        h = F.relu(F.linear(h, P["fc7_w"], P["fc7_b"]))
        logits = F.linear(h, P["cls_w"], P["cls_b"]).numpy()
        deltas = F.linear(h, P["bbox_w"], P["bbox_b"]).numpy() * np.tile(np.asarray(P["bbox_std"], np.float32), C) + np.tile(np.asarray(P["bbox_mean"], np.float32), C)
    cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
    assert np.abs(cls[idx] - logits).max() < 1e-4 and np.abs(raw[idx] - deltas).max() < 1e-4


@pytest.fixture(scope="module")
def cpu_feats(O, dev, full):
    """CPU-side quantities the trained-scale tests share (fc6 / fc7 are common to every head regime):
      feat_o / feat_t / feat_d   conv5 of the oracle (plain C fp32), of PyTorch-CPU (oneDNN fp32) and of the DEVICE;
      fc7_t                      PyTorch-CPU fp32 fc7 activations of all 1000 ROIs (on its own conv5);
      idx, fc7_64[src]           a 100-ROI sample and, for each conv5 source, fc7 of the sample evaluated in FLOAT64 — the exact head,
                                 against which an fp32 head's own summation error can be separated from the trunk's."""
    import torch.nn.functional as F
    from multipathnet_amd import models
    im, boxes, P, net = full["im"], full["boxes"], full["P"], full["net"]
    Pn = _np_tree(P)
    feat_o = O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"])
    net.detect(full["imd"], full["bd"])
    feat_d = net.debug_tensor("conv5", feat_o.shape).cpu().numpy()
    rois = O.project_im_rois(boxes, 1.0)
    idx = np.random.default_rng(7).choice(boxes.shape[0], N_SAMPLE, replace=False)
    with torch.no_grad():
        x = torch.from_numpy(O.image_transform(im, **O.ROSS)).unsqueeze(0)
        li = 0
        for item in models.VGG16_CFG:
            if item == "P":
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            else:
                x = F.relu(F.conv2d(x, P["conv_w"][li], P["conv_b"][li], padding=1))
                li += 1
        feat_t = x[0].numpy()
        pooled_t, _ = O.roi_pool(feat_t, rois, 7, 7, 1.0 / 16)
        h = torch.from_numpy(pooled_t.reshape(boxes.shape[0], -1))
        h = F.relu(F.linear(h, P["fc6_w"], P["fc6_b"]))
        fc7_t = F.relu(F.linear(h, P["fc7_w"], P["fc7_b"]))
        w6, b6, w7, b7 = P["fc6_w"].double(), P["fc6_b"].double(), P["fc7_w"].double(), P["fc7_b"].double()
        fc7_64 = {}
        for name, feat in (("oracle", feat_o), ("torch", feat_t), ("device", feat_d)):
            pooled, _ = O.roi_pool(feat, rois[idx], 7, 7, 1.0 / 16)   # integer bin bounds + max: no arithmetic
            h = torch.from_numpy(pooled.reshape(N_SAMPLE, -1)).double()
            fc7_64[name] = F.relu(F.linear(F.relu(F.linear(h, w6, b6)), w7, b7))
    return dict(feat=feat_o, feat_t=feat_t, feat_d=feat_d, fc7=fc7_t, idx=idx, fc7_64=fc7_64)


def _regime_params(full, cpu_feats, regime):
    """'trained': random head weights / biases at a trained detector's magnitude (models.synthetic_params(head_scale=...));
    'saturated': conftest.saturated_heads — fitted so that nearly every softmax row saturates to exactly 1.0f and every
    foreground class holds several ROIs tied at 1.0f."""
    import bench
    from conftest import saturated_heads
    from multipathnet_amd import models
    if regime == "trained":
        Q = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557, head_scale="trained")
        assert torch.equal(Q["fc6_w"], full["P"]["fc6_w"]) and torch.equal(Q["conv_w"][5], full["P"]["conv_w"][5])
        return Q
    return saturated_heads(full["P"], cpu_feats["fc7"], full["boxes"], bench.N_CLASSES)


@pytest.fixture(scope="module", params=["trained", "saturated"])
def scaled(request, O, dev, full, cpu_feats):
    import bench
    from multipathnet_amd import models
    Q = _regime_params(full, cpu_feats, request.param)
    net = models.FastRCNN(Q, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    return dict(regime=request.param, P=Q, net=net)


def test_fullsize_trained_scale_logits_deltas_vs_oracle(O, dev, full, cpu_feats, scaled):
    """the full-size check at the score scale of a trained detector — head weights 3-9x the initialisation,
    non-zero biases, logits of +-10..16, softmax rows saturating to exactly 1.0f.

    'trained' (random heads, |cls_w| rms 0.03): class LOGITS and box DELTAS within 1e-4 ABSOLUTE of the oracle, like the
    initialisation-scale test above.

    'saturated' (heads fitted to this image's features, |cls_w| rms 0.09, every softmax row's winner exactly 1.0f): north_star's
    quantities — class SCORES and bbox regression — are within 1e-4; the pre-softmax logits are not, for ANY fp32 implementation:
    the fitted weights weigh the low-variance directions of fc7 and amplify fp32 summation-order error ~10x.  Measured on CPU
    alone (tools/logit_error_attribution.py): PyTorch-CPU fp32 and the oracle's plain-C fp32 differ by 7.9e-4 there, the oracle's own
    head is 6.3e-4 from the same head evaluated in float64 (sequential accumulation over K = 25088), PyTorch's blocked summation
    1.1e-4.  So the logit bound in this regime is what fp32 admits, stated relative to the two CPU implementations:
      (i) device-vs-oracle no larger than 1.5x PyTorch-vs-oracle;
      (ii) the device's HEAD, against the float64 head on the device's own conv5, at least as accurate as the oracle's head is
           against the float64 head on the oracle's conv5;
      (iii) the device's TRUNK, seen through the float64 head, no further from the oracle's trunk than 1.5x PyTorch's trunk is.
    The three error terms are printed: this is the per-layer attribution VERDICT r2 asked for."""
    import torch.nn.functional as F
    net, P, regime = scaled["net"], scaled["P"], scaled["regime"]
    im, boxes = full["im"], full["boxes"]
    s, b = net.detect(full["imd"], full["bd"])
    s, b = s.cpu().numpy(), b.cpu().numpy()
    N, C = boxes.shape[0], net.n_classes
    cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
    Pn = _np_tree(P)
    idx = cpu_feats["idx"]
    logits, deltas = O.frcnn_head(cpu_feats["feat"], O.project_im_rois(boxes[idx], 1.0), Pn)
    e_logit, e_delta = np.abs(cls[idx] - logits).max(), np.abs(raw[idx] - deltas).max()
    with torch.no_grad():
        logits_t = F.linear(cpu_feats["fc7"][idx], P["cls_w"], P["cls_b"]).numpy()
        l64 = {k: F.linear(v, P["cls_w"].double(), P["cls_b"].double()).numpy() for k, v in cpu_feats["fc7_64"].items()}
    e_ref = np.abs(logits_t - logits).max()                       # two CPU fp32 implementations of the same network
    head_dev, head_orc, head_torch = np.abs(cls[idx] - l64["device"]).max(), np.abs(logits - l64["oracle"]).max(), np.abs(logits_t - l64["torch"]).max()
    trunk_dev, trunk_torch = np.abs(l64["device"] - l64["oracle"]).max(), np.abs(l64["torch"] - l64["oracle"]).max()
    n_sat = int((s == 1.0).sum())
    print("full-size head [%s], %d ROIs: max|dlogit| device-oracle = %.3g, PyTorchCPU-oracle = %.3g (logits %.3g .. %.3g, |cls_w| rms %.3g); "
          "head error vs the float64 head on the same conv5: device %.3g, oracle %.3g, PyTorchCPU %.3g; trunk difference to the oracle's through "
          "the float64 head: device %.3g, PyTorchCPU %.3g; max|ddelta| = %.3g (|delta| <= %.3g); %d of %d softmax rows hold an exact 1.0f (%d foreground)"
          % (regime, N_SAMPLE, e_logit, e_ref, cls.min(), cls.max(), float(np.sqrt((Pn["cls_w"] ** 2).mean())), head_dev, head_orc, head_torch,
             trunk_dev, trunk_torch, e_delta, np.abs(deltas).max(), n_sat, N, int((s[:, 1:] == 1.0).sum())))
    assert np.abs(cls).max() > 9.0                      # the regime is what it claims to be
    assert e_delta < 1e-4                               # ABSOLUTE, pre-decode
    if regime == "trained":
        assert e_logit < 1e-4                           # ABSOLUTE, pre-softmax
    else:
        assert n_sat >= 300 and int((s[:, 1:] == 1.0).sum()) >= 20
        assert e_logit < max(1e-4, 1.5 * e_ref)         # (i)
    assert head_dev <= max(2e-5, head_orc)              # (ii)
    assert trunk_dev <= max(2e-5, 1.5 * trunk_torch)    # (iii)
    assert np.abs(s[idx] - O.softmax(logits)).max() < 1e-4   # north_star: class scores
    bo = O.clamp_boxes(O.bbox_decode(boxes[idx], deltas), im.shape[2], im.shape[1])
    assert np.abs(b[idx] - bo).max() < 1e-4 * im.shape[2]


def test_fullsize_trained_scale_vs_pytorch_cpu(O, dev, full, cpu_feats, scaled):
    """The same regimes against PyTorch-CPU's fp32 path (oneDNN; oracle-independent dense arithmetic) on ALL 1000 ROIs: scores and
    deltas within 1e-4; logits within 1e-4 ('trained') / within 1.5x the PyTorch-vs-oracle spread of the sample ('saturated', see
    the test above)."""
    import torch.nn.functional as F
    net, P = scaled["net"], scaled["P"]
    N, C = full["boxes"].shape[0], net.n_classes
    s, _ = net.detect(full["imd"], full["bd"])
    with torch.no_grad():
        logits = F.linear(cpu_feats["fc7"], P["cls_w"], P["cls_b"]).numpy()
        deltas = F.linear(cpu_feats["fc7"], P["bbox_w"], P["bbox_b"]).numpy() * np.tile(np.asarray(P["bbox_std"], np.float32), C) + np.tile(np.asarray(P["bbox_mean"], np.float32), C)
    cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
    e_logit = np.abs(cls - logits).max()
    print("[%s] vs PyTorch-CPU, 1000 ROIs: max|dlogit| = %.3g, max|ddelta| = %.3g" % (scaled["regime"], e_logit, np.abs(raw - deltas).max()))
    assert np.abs(raw - deltas).max() < 1e-4 and np.abs(s.cpu().numpy() - O.softmax(logits)).max() < 1e-4
    if scaled["regime"] == "trained":
        assert e_logit < 1e-4
    else:
        idx = cpu_feats["idx"]
        lo, _ = O.frcnn_head(cpu_feats["feat"], O.project_im_rois(full["boxes"][idx], 1.0), _np_tree(P))
        assert e_logit < max(1e-4, 1.5 * np.abs(logits[idx] - lo).max())


def test_fullsize_trained_scale_test_one_all_classes_vs_reference_nms(O, dev, full, scaled):
    """Tester:testOne at full size in the trained / saturated regimes: for ALL 20 foreground classes the device's per-class NMS
    equals the reference's own nms.c on the device's scored rows, bit for bit (boxes, order, source indices) — the saturated
    regime's classes hold several scores tied at exactly 1.0f, so the tie dispatch runs inside the pipeline here, not only in
    the micro-tests — and the top-100 record follows from those tables by the keep_top_k rule."""
    net, regime = scaled["net"], scaled["regime"]
    imd, bd = full["imd"], full["bd"]
    s, b = net.detect(imd, bd)
    dets, n = net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    keep, kidx, nk = [t.cpu().numpy() for t in net.nms_results()]
    sn, bn = s.cpu().numpy(), b.cpu().numpy()
    C = net.n_classes
    per, tied = [], 0
    for cls in range(1, C):
        sb, src = O.select_scored(sn, bn, cls, -1.5)
        tied += int(np.unique(sb[:, 4]).size < sb.shape[0])
        ref = O.ref_nms(sb, 0.3)
        mine, ridx = O.nms(sb, 0.3, return_index=True)
        assert np.array_equal(mine, ref)
        k = int(nk[cls - 1])
        assert k == ref.shape[0] and np.array_equal(keep[cls - 1, :k], ref), cls
        assert np.array_equal(kidx[cls - 1, :k], src[ridx]), cls
        per.append(ref)
    print("[%s] classes with bit-equal scores: %d of %d" % (regime, tied, C - 1))
    if regime == "saturated":
        assert tied == C - 1
    kept, _ = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k, np.full((k.shape[0], 1), j + 1, np.float32)], 1) for j, k in enumerate(kept) if k.size])
    assert np.array_equal(dets[: int(n.item())].cpu().numpy(), exp)


def iterloc_vote_check(O, dev, plain_net, fused_net, imd, bd, voting, score_pow, label):
    """Tester_FRCNN.lua:82-99,118-124 at FULL size: opt.test_num_iterative_loc = 2 [+ test_bbox_voting] in the fused device
    path `fused_net`; the joined 2N-row score / box tables come from the host mirror (detect.Tester_FRCNN on `plain_net`, same kernels).
    Every foreground class's NMS over its 2N-row table must equal the reference's COMPILED nms.c (O.ref_nms) — kept count, source indices —
    and the voted rows the compiled bbox_vote (nms.c:110-142, O.ref_bbox_vote) bit for bit, with scores:pow(p) as THFloatTensor_pow does it;
    the top-100 record follows by utils.keep_top_k.  Returns (tied classes, rows moved by voting) for the caller's regime assertions."""
    from multipathnet_amd import detect
    N, C = bd.shape[0], plain_net.n_classes
    dets, n = fused_net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    keep, kidx, nk = [t.cpu().numpy() for t in fused_net.nms_results()]
    assert keep.shape[1] == 2 * N
    tester = detect.Tester_FRCNN(plain_net, opt={"test_num_iterative_loc": 2})
    _, (output, bbox_pred) = tester.testOne(imd, bd)
    sc, bb = output.cpu().numpy(), bbox_pred.cpu().numpy()
    assert sc.shape == (2 * N, C) and bb.shape == (2 * N, 4 * C)
    # the first pass is clamped to the image (Tester_FRCNN.lua:75-78), the refinement pass is not
    assert bb[:N].min() >= 1.0
    per, tied, moved, widest = [], 0, 0, 0
    for cls in range(1, C):
        sb, src = O.select_scored(sc, bb, cls, -1.5)
        assert sb.shape[0] == 2 * N                         # thresh = -1.5: every row of both passes (Tester_FRCNN.lua:50)
        tied += int(np.unique(sb[:, 4]).size < sb.shape[0])
        ref = O.ref_nms(sb, 0.3)
        mine, ridx = O.nms(sb, 0.3, return_index=True)
        assert np.array_equal(mine, ref)
        k = int(nk[cls - 1])
        assert k == ref.shape[0], (label, cls, k, ref.shape[0])
        assert np.array_equal(kidx[cls - 1, :k], src[ridx]), (label, cls)
        widest = max(widest, k)
        if voting:
            votes = sb.copy()
            if score_pow != 1.0:
                votes[:, 4] = np.power(votes[:, 4].astype(np.float64), float(np.float32(score_pow))).astype(np.float32)
            voted = O.ref_bbox_vote(ref, votes, 0.5)
            assert np.array_equal(voted[:, 4], ref[:, 4])   # the kept boxes keep their NMS scores (nms.c:139)
            moved += int((voted[:, :4] != ref[:, :4]).any(1).sum())
            ref = voted
        assert np.array_equal(keep[cls - 1, :k], ref, equal_nan=True), (label, cls)
        per.append(ref)
    kept, _ = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k_, np.full((k_.shape[0], 1), j + 1, np.float32)], 1) for j, k_ in enumerate(kept) if k_.size])
    nd = int(n.item())
    assert nd == exp.shape[0] and np.array_equal(dets[:nd].cpu().numpy(), exp, equal_nan=True)
    print("[%s] num_iter=2 voting=%s pow=%g: %d classes x %d rows, %d with bit-equal scores, widest kept table %d, %d rows moved by the vote"
          % (label, voting, score_pow, C - 1, 2 * N, tied, widest, moved))
    return tied, moved


@pytest.mark.parametrize("voting,score_pow", [(False, 1.0), (True, 1.0), (True, 0.5)])
def test_fullsize_iterative_localisation_and_voting_vs_compiled_reference(O, dev, full, scaled, voting, score_pow):
    """configs[1] with the reference's accuracy knobs on, 600x1000 x 1000 ROIs -> 2000-row class tables, in the trained and the saturated
    score regime (the latter: every class holds scores tied at exactly 1.0f — the wide-table tie paths and bbox_vote_batched run on them)"""
    import bench
    from multipathnet_amd import models
    net = models.FastRCNN(scaled["P"], max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS, num_iter=2, bbox_voting=voting,
                          bbox_vote_thresh=0.5, bbox_vote_score_pow=score_pow)
    tied, moved = iterloc_vote_check(O, dev, scaled["net"], net, full["imd"], full["bd"], voting, score_pow, "vgg16-frcnn/" + scaled["regime"])
    if scaled["regime"] == "saturated":
        assert tied == net.n_classes - 1
    if voting:
        assert moved > 0
    del net
    torch.cuda.empty_cache()


def test_fullsize_2000_proposals(O, dev, full, cpu_feats):
    """VGG-16 Fast R-CNN at N = 2000 — the proposal count the reference's own evaluation script uses (scripts/eval_fastrcnn_voc2007.sh:8-20:
    -test_best_proposals_number 2000) — on the 600x1000 image with trained-scale heads: (1) class logits / box deltas of a 100-ROI sample
    (drawn over all 2000 rows, incl. the last) within 1e-4 absolute of the oracle; (2) the first 1000 rows BIT-EQUAL to the 1000-ROI
    pipeline's (rows do not depend on the batch: one fc6 GEMM over 2000 rows vs 1000); (3) all 20 classes' NMS over 2000-row tables ==
    the reference's compiled nms.c, and the top-100 record by utils.keep_top_k."""
    import bench
    from multipathnet_amd import models
    N = 2000
    Q = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557, head_scale="trained")
    boxes = bench.more_boxes(full["boxes"], N)
    assert np.array_equal(boxes[: bench.N_ROIS], full["boxes"])
    net = models.FastRCNN(Q, max_h=bench.H, max_w=bench.W, max_rois=N)
    bd = torch.from_numpy(boxes).to(dev)
    s, b = net.detect(full["imd"], bd)
    C = net.n_classes
    cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
    raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
    idx = np.random.default_rng(2000).choice(N, N_SAMPLE, replace=False)
    idx[0] = N - 1
    logits, deltas = O.frcnn_head(cpu_feats["feat"], O.project_im_rois(boxes[idx], 1.0), _np_tree(Q))
    e_l, e_d = np.abs(cls[idx] - logits).max(), np.abs(raw[idx] - deltas).max()
    print("N = 2000, %d-ROI sample: max|dlogit| = %.3g (logits %.3g .. %.3g), max|ddelta| = %.3g" % (N_SAMPLE, e_l, cls.min(), cls.max(), e_d))
    assert np.abs(cls).max() > 9.0 and e_l < 1e-4 and e_d < 1e-4
    net1k = models.FastRCNN(Q, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    s1, b1 = net1k.detect(full["imd"], full["bd"])
    assert torch.equal(s[: bench.N_ROIS], s1) and torch.equal(b[: bench.N_ROIS], b1)
    del net1k
    dets, n = net.test_one_async(full["imd"], bd)
    torch.cuda.synchronize()
    keep, kidx, nk = [t.cpu().numpy() for t in net.nms_results()]
    sn, bn = s.cpu().numpy(), b.cpu().numpy()
    per = []
    for c in range(1, C):
        sb, src = O.select_scored(sn, bn, c, -1.5)
        ref = O.ref_nms(sb, 0.3)
        mine, ridx = O.nms(sb, 0.3, return_index=True)
        k = int(nk[c - 1])
        assert np.array_equal(mine, ref) and k == ref.shape[0] and np.array_equal(keep[c - 1, :k], ref), c
        assert np.array_equal(kidx[c - 1, :k], src[ridx]), c
        per.append(ref)
    kept, _ = O.keep_top_k(per, 100)
    exp = np.concatenate([np.concatenate([k_, np.full((k_.shape[0], 1), j + 1, np.float32)], 1) for j, k_ in enumerate(kept) if k_.size])
    assert np.array_equal(dets[: int(n.item())].cpu().numpy(), exp)
    del net
    torch.cuda.empty_cache()


def test_fullsize_mixed_size_stream(O, dev, full):
    """A dataset-shaped stream at full size (VERDICT r4 missing #3): Tester:test feeds images of DIFFERENT sizes (Tester_FRCNN.lua:150-157), each
    rescaled by getImages to a 600-px short side / 1000-px cap (ImageDetect.lua:22-52) — bench.MIXED_SIZES: s = 1 at two aspect ratios, s = 1.25
    landscape and portrait, the capped scale, and a 2x downscale; ragged proposal counts.  (1) one ROI sample per size against the oracle's whole
    path (transform -> image.scale -> trunk -> head; logits / deltas 1e-4 absolute); (2) the host-fed pipelined form (what bench.py times: uploads
    on the copy stream, tails on the side stream, activation halos re-laid at every size change) over 3 rounds of the stream == the serial
    records bit for bit."""
    import bench
    from multipathnet_amd import models
    P = full["P"]
    net = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=bench.N_ROIS, scale=600, max_size=1000)
    stream = bench.mixed_size_inputs()
    Pn = _np_tree(P)
    ref = []
    for im, bx in stream:
        imd, bd = torch.from_numpy(im).to(dev), torch.from_numpy(bx).to(dev)
        H0, W0 = im.shape[1:]
        sc = O.pick_scale(H0, W0, 600, 1000)
        s, b = net.detect(imd, bd)
        N, C = bx.shape[0], net.n_classes
        cls = net.debug_tensor("cls", (N, C)).cpu().numpy()
        raw = net.debug_tensor("bbox_raw", (N, 4 * C)).cpu().numpy()
        idx = np.random.default_rng(H0).choice(N, 40, replace=False)
        idx[0] = N - 1
        so, bo, logits, deltas = O.detect(im, bx[idx], Pn, target=600, max_size=1000)
        e_l, e_d = np.abs(cls[idx] - logits).max(), np.abs(raw[idx] - deltas).max()
        print("%4d x %4d (s = %.4g -> %d x %d), %4d ROIs: max|dlogit| = %.3g, max|ddelta| = %.3g" % (H0, W0, sc, int(H0 * sc), int(W0 * sc), N, e_l, e_d))
        assert e_l < 1e-4 and e_d < 1e-4
        assert np.abs(s.cpu().numpy()[idx] - so).max() < 1e-4
        assert np.abs(b.cpu().numpy()[idx] - O.clamp_boxes(bo, W0, H0)).max() < 1e-4 * max(H0, W0)
        d, n = net.test_one_async(imd, bd)
        torch.cuda.synchronize()
        ref.append(d[: int(n.item())].clone())
        assert ref[-1].shape[0] > 0
    pin = [(torch.from_numpy(im).pin_memory(), torch.from_numpy(bx).pin_memory()) for im, bx in stream]
    steps = 3 * len(stream)
    outs = [(torch.zeros_like(net._dets), torch.zeros_like(net._n_dets)) for _ in range(steps)]
    from multipathnet_amd._lib import check, f32p
    from multipathnet_amd.nn import _f, _i, _stream
    import ctypes as C_
    order = [t % len(stream) for t in range(steps)]
    order[-3:] = [0, 0, 5]     # also: the same size twice in a row, then the largest upload last
    for t in range(steps):
        i, bx = pin[order[t]]
        d, n = outs[t]
        check(net._lib.mpn_frcnn_test_one_pipelined_host(net._h, C_.cast(i.data_ptr(), f32p), i.shape[1], i.shape[2], C_.cast(bx.data_ptr(), f32p),
                                                         bx.size(0), _f(d), d.size(0), _i(n), _stream()), "pipelined_host")
    net.flush()
    torch.cuda.synchronize()
    for t in range(steps):
        d, n = outs[t]
        assert torch.equal(d[: int(n.item())], ref[order[t]]), (t, order[t])
    del net
    torch.cuda.empty_cache()
    # a size change re-lays only the HALOS of the activations (c8p_zero_halos_kernel, round 5); clearing every buffer whole, as rounds 1-4
    # did (debug hook), must give the same records — in an order in which every size follows a LARGER and a smaller one
    from conftest import hooks
    with hooks(halo_memset=1):
        net2 = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=bench.N_ROIS, scale=600, max_size=1000)
        for i in (5, 0, 4, 3, 2, 1, 0, 5, 3):
            im, bx = stream[i]
            d, n = net2.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(bx).to(dev))
            torch.cuda.synchronize()
            assert torch.equal(d[: int(n.item())], ref[i]), i
        del net2
    net3 = models.FastRCNN(P, max_h=1000, max_w=1000, max_rois=bench.N_ROIS, scale=600, max_size=1000)
    for i in (5, 0, 4, 3, 2, 1, 0, 5, 3):
        im, bx = stream[i]
        d, n = net3.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(bx).to(dev))
        torch.cuda.synchronize()
        assert torch.equal(d[: int(n.item())], ref[i]), i
    del net3
    torch.cuda.empty_cache()


def test_fullsize_pipelined_host_record_equals_serial(dev, full):
    """what bench.py times (mpn_frcnn_test_one_pipelined_host at 600x1000 x 1000 ROIs: upload on the copy
    stream, NMS / top-k tail on the side stream) returns, image after image, exactly the record the serial
    mpn_frcnn_test_one returns — three different images in rotation, 12 steps, no host sync inside the loop."""
    import bench
    net = full["net"]
    rng = np.random.default_rng(99)
    ims = [full["im"]] + [rng.random((3, bench.H, bench.W), dtype=np.float32) for _ in range(2)]
    bxs = [full["boxes"], full["boxes"][::-1].copy(), full["boxes"][rng.permutation(bench.N_ROIS)].copy()]
    ref = []
    for im, bx in zip(ims, bxs):
        d, n = net.test_one_async(torch.from_numpy(im).to(dev), torch.from_numpy(bx).to(dev))
        torch.cuda.synchronize()
        ref.append(d[: int(n.item())].clone())
    assert not torch.equal(ref[0], ref[1]) and not torch.equal(ref[1], ref[2])
    pin = [(torch.from_numpy(im).pin_memory(), torch.from_numpy(bx).pin_memory()) for im, bx in zip(ims, bxs)]
    steps = 12
    outs = [(torch.zeros_like(net._dets), torch.zeros_like(net._n_dets)) for _ in range(steps)]
    from multipathnet_amd._lib import check, f32p
    from multipathnet_amd.nn import _f, _i, _stream
    import ctypes as C
    for t in range(steps):
        i, bx = pin[t % 3]
        d, n = outs[t]
        check(net._lib.mpn_frcnn_test_one_pipelined_host(net._h, C.cast(i.data_ptr(), f32p), bench.H, bench.W, C.cast(bx.data_ptr(), f32p),
                                                         bx.size(0), _f(d), d.size(0), _i(n), _stream()), "pipelined_host")
    net.flush()
    torch.cuda.synchronize()
    for t in range(steps):
        d, n = outs[t]
        assert torch.equal(d[: int(n.item())], ref[t % 3]), t


def test_fullsize_properties(O, dev, full):
    net, imd, bd, boxes = full["net"], full["imd"], full["bd"], full["boxes"]
    N, C = boxes.shape[0], net.n_classes
    s, b = net.detect(imd, bd)
    # softmax rows, clamping (Tester_FRCNN.lua:75-78)
    assert float((s.sum(1) - 1).abs().max()) < 1e-5
    bb = b.view(N, C, 4)
    assert float(bb[..., 0::2].min()) >= 1 and float(bb[..., 0::2].max()) <= imd.shape[2]
    assert float(bb[..., 1::2].min()) >= 1 and float(bb[..., 1::2].max()) <= imd.shape[1]
    # determinism and ROI-order equivariance (rows are independent: memoryEfficientForward's chunk invariance, ImageDetect.lua:126-133)
    s2, b2 = net.detect(imd, bd)
    assert torch.equal(s, s2) and torch.equal(b, b2)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(N)).to(dev)
    sp, bp = net.detect(imd, bd[perm].contiguous())
    assert torch.equal(sp, s[perm]) and torch.equal(bp, b[perm])
    # cached trunk features (recompute_features = false) == full run
    sc, bc = net.detect(imd, bd, recompute_features=False)
    assert torch.equal(sc, s) and torch.equal(bc, b)
    # Tester:testOne: per-class NMS on the device == the reference's own nms.c on the same scored boxes, class by class
    net.test_one_async(imd, bd)
    torch.cuda.synchronize()
    keep, kidx, nk = net.nms_results()
    keep, nk = keep.cpu().numpy(), nk.cpu().numpy()
    sn, bn = s.cpu().numpy(), b.cpu().numpy()
    for cls in (1, 7, 20):
        sb = np.concatenate([bn[:, 4 * cls:4 * cls + 4], sn[:, cls:cls + 1]], 1).astype(np.float32)
        ref = O.ref_nms(sb, 0.3)
        k = int(nk[cls - 1])
        assert k == ref.shape[0] and np.array_equal(keep[cls - 1, :k], ref)
        # idempotence: NMS of the kept set keeps everything, in the same order
        assert np.array_equal(O.ref_nms(ref, 0.3), ref)
    # keep_top_k(100): at most top_k rows unless ties at the threshold, scores >= the k-th largest kept score
    n = int(net._n_dets.item())
    dets = net._dets[:n].cpu().numpy()
    allk = np.concatenate([keep[c, :int(nk[c]), 4] for c in range(C - 1)])
    thr = np.sort(allk)[::-1][min(100, allk.size) - 1]
    assert n == int((allk >= thr).sum()) and dets[:, 4].min() >= thr


def test_multipathnet_fullsize_scores_vs_oracle_on_roi_sample(O, dev):
    """BASELINE configs[2] at full size (VGG-16 MultiPathNet: 4 foveal towers + box tower with conv3/4/5 skip pooling, K = 6
    integral classifiers, 81 classes, 1000 ROIs on the 600x1000 image — what tools/bench_mpn.py times): the oracle runs the
    trunk with its conv3 / conv4 / conv5 taps and the head for 24 of the 1000 ROIs (rows are independent); the bbox-regression
    deltas (pre-decode) are compared too, absolute 1e-4."""
    import bench
    from multipathnet_amd import models
    P = models.synthetic_mpnet_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=81, n_integral=6, seed=557)
    net = models.MultiPathNet(P, max_h=bench.H, max_w=bench.W, max_rois=bench.N_ROIS)
    im, boxes = bench.synthetic_inputs()
    scores, bbox = net.detect(torch.from_numpy(im).to(dev), torch.from_numpy(boxes).to(dev))
    scores, bbox = scores.cpu().numpy(), bbox.cpu().numpy()
    assert np.abs(scores.sum(1) - 1).max() < 1e-5
    def tree(v):
        if isinstance(v, dict):
            return {k: tree(x) for k, x in v.items()}
        if isinstance(v, list):
            return [tree(x) for x in v]
        return v.numpy() if hasattr(v, "numpy") else v
    Pn = tree(P)
    taps = {}
    O.vgg_trunk(O.image_transform(im, **O.ROSS), Pn["conv_w"], Pn["conv_b"], taps=taps)
    idx = np.random.default_rng(17).choice(boxes.shape[0], 24, replace=False)
    ref_scores, deltas = O.mpnet_head([taps["conv5"], taps["conv4"], taps["conv3"]], O.project_im_rois(boxes[idx], 1.0), Pn)
    raw = net.debug_tensor("bbox_raw", (boxes.shape[0], 4 * 81)).cpu().numpy()
    assert np.abs(raw[idx] - deltas).max() < 1e-4
    ref_bbox = O.clamp_boxes(O.bbox_decode(boxes[idx], deltas), im.shape[2], im.shape[1])
    assert np.abs(scores[idx] - ref_scores).max() < 1e-4
    assert np.abs(bbox[idx] - ref_bbox).max() < 1e-4 * im.shape[2]


@pytest.mark.parametrize("n", [1000, 2000])
def test_fullsize_fc_three_plane_split_beside_the_fp32_pipeline(O, dev, full, n):
    """MPN_FC_SPLIT3 (fc6 / fc7 as exact three-plane bf16 splits, six bf16 MFMA products per k-step, fp32 accumulate) at FULL size with trained-scale
    heads, isolated from the trunk: both pipelines pool the same device conv5 map, so the difference of their logits / deltas is the fc
    arithmetic alone.  Reported against a float64 head (fc6 -> fc7 -> cls / bbox in float64 on the device's own pooled operand of a 100-ROI
    sample): the split must be no further from it than the fp32 MFMA pipeline is (x 1.2), and the two pipelines within 5e-5 of each other on
    logits of magnitude ~16 (the 1e-4 budget of the path's parity gate is spent on the trunk: tests above)."""
    import bench
    from multipathnet_amd import models
    from conftest import hooks
    Q = models.synthetic_params(models.VGG16_CFG, pooled=7, fc_dim=4096, n_classes=bench.N_CLASSES, seed=557, head_scale="trained")
    boxes = bench.more_boxes(full["boxes"], n)
    bd = torch.from_numpy(boxes).to(dev)
    C = bench.N_CLASSES
    out = {}
    for arith in ("fp32", "split3"):
        net = models.FastRCNN(Q, max_h=bench.H, max_w=bench.W, max_rois=n, fc_arith=arith)
        net.detect(full["imd"], bd)
        out[arith] = (net.debug_tensor("cls", (n, C)).cpu().numpy(), net.debug_tensor("bbox_raw", (n, 4 * C)).cpu().numpy())
        if arith == "fp32":
            idx = np.random.default_rng(3).choice(n, N_SAMPLE, replace=False)
            pooled = net.debug_tensor("pooled", (n, 512, 7, 7))[torch.from_numpy(idx).to(dev)].cpu().numpy().reshape(N_SAMPLE, -1).astype(np.float64)
        del net
        torch.cuda.empty_cache()
    Qn = _np_tree(Q)
    h6 = np.maximum(pooled @ Qn["fc6_w"].astype(np.float64).T + Qn["fc6_b"], 0)
    h7 = np.maximum(h6 @ Qn["fc7_w"].astype(np.float64).T + Qn["fc7_b"], 0)
    l64 = h7 @ Qn["cls_w"].astype(np.float64).T + Qn["cls_b"]
    e32, e3 = np.abs(out["fp32"][0][idx] - l64).max(), np.abs(out["split3"][0][idx] - l64).max()
    d_l, d_d = np.abs(out["fp32"][0] - out["split3"][0]).max(), np.abs(out["fp32"][1] - out["split3"][1]).max()
    bias = float((out["split3"][0].astype(np.float64) - out["fp32"][0]).mean())
    print("N = %d, logits up to %.3g: vs a float64 head on the same pooled operand (%d ROIs): fp32 MFMA %.3g, three-plane split %.3g; split - fp32 pipeline: "
          "max |dlogit| %.3g (mean signed %.2g), max |ddelta| %.3g" % (n, np.abs(out["fp32"][0]).max(), N_SAMPLE, e32, e3, d_l, bias, d_d))
    assert e3 < max(1.2 * e32, 1e-5)          # measured: 1.3-1.4e-5 against the fp32 MFMA pipeline's 2.0-2.1e-5 (two accumulator sets: dense.hip)
    assert d_l < 5e-5 and d_d < 3e-6          # the two pipelines differ by no more than their two distances to the float64 head add up to
