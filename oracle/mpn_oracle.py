"""numpy/ctypes front-end of the CPU oracle (oracle/mpn_oracle.c, oracle/_ref/libnms_ref.so).

TEST INFRASTRUCTURE ONLY.  Importable solely from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never from ``multipathnet_amd`` (the product fails loudly when
its HIP library is missing; it has no CPU fallback).

Each function forwards to the C restatement named ``orc_<name>``; see mpn_oracle.c for the
reference file:line every one follows.  ``frcnn_forward`` composes them into the per-image path
of Tester_FRCNN.lua:54-139 / ImageDetect.lua:156-193 / models/vgg.lua:23-31.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile libmpn_oracle.so (and _ref/libnms_ref.so when /root/reference is present)."""
    so = os.path.join(_HERE, "libmpn_oracle.so")
    src = os.path.join(_HERE, "mpn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmpn_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libnms_ref.so")
    if os.path.exists("/root/reference/nms.c") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _p(a, t=f32p):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(os.path.join(_HERE, "libmpn_oracle.so"))
        _LIB.orc_overlap.restype = C.c_float
        _LIB.orc_nms.restype = C.c_int
        _LIB.orc_select_scored.restype = C.c_int
        _LIB.orc_topk_threshold.restype = C.c_float
        _LIB.orc_pick_scale.restype = C.c_double
    return _LIB


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libnms_ref.so"))


def ref():
    """The reference's own nms.c (compiled unmodified) — None-safe: raises if absent."""
    global _REF
    if _REF is None:
        build()
        _REF = C.CDLL(os.path.join(_HERE, "_ref", "libnms_ref.so"))
        _REF.overlap.restype = C.c_float
        _REF.mpn_th_shim_new.restype = C.c_void_p
        _REF.mpn_th_shim_from.restype = C.c_void_p
        _REF.mpn_th_shim_from.argtypes = [f32p, C.c_long, C.c_long]
        _REF.THFloatTensor_data.restype = f32p
        _REF.THFloatTensor_data.argtypes = [C.c_void_p]
        _REF.mpn_th_shim_size.restype = C.c_long
        _REF.mpn_th_shim_size.argtypes = [C.c_void_p, C.c_int]
        _REF.mpn_th_shim_free.argtypes = [C.c_void_p]
        _REF.NMS.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _REF.bbox_vote.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    return _REF


# ---------------------------------------------------------------- reference nms.c (real code)
def _th_to_np(r, t):
    n0, n1 = r.mpn_th_shim_size(t, 0), r.mpn_th_shim_size(t, 1)
    if n0 * n1 == 0:
        return np.zeros((n0, 5), np.float32)
    return np.ctypeslib.as_array(r.THFloatTensor_data(t), shape=(n0, n1)).copy()


def ref_overlap(a, b):
    a, b = _f32(a), _f32(b)
    return float(ref().overlap(_p(a), _p(b)))


def ref_nms(scored_boxes, thr):
    r = ref()
    sb = _f32(scored_boxes).reshape(-1, 5)
    t_in = r.mpn_th_shim_from(_p(sb), sb.shape[0], 5)
    t_out = r.mpn_th_shim_new()
    r.NMS(t_out, t_in, C.c_float(thr))
    out = _th_to_np(r, t_out) if sb.shape[0] else np.zeros((0, 5), np.float32)
    r.mpn_th_shim_free(t_in)
    r.mpn_th_shim_free(t_out)
    return out


def ref_bbox_vote(nms_boxes, scored_boxes, thr):
    r = ref()
    nb, sb = _f32(nms_boxes).reshape(-1, 5), _f32(scored_boxes).reshape(-1, 5)
    t_n = r.mpn_th_shim_from(_p(nb), nb.shape[0], 5)
    t_s = r.mpn_th_shim_from(_p(sb), sb.shape[0], 5)
    t_o = r.mpn_th_shim_new()
    r.bbox_vote(t_o, t_n, t_s, C.c_float(thr))
    out = _th_to_np(r, t_o) if nb.shape[0] else np.zeros((0, 5), np.float32)
    for t in (t_n, t_s, t_o):
        r.mpn_th_shim_free(t)
    return out


# ---------------------------------------------------------------- restated ops
def overlap(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_overlap(_p(a), _p(b)))


def boxoverlap(a, b):
    a, b = _f32(a).reshape(-1, 4), _f32(b)
    out = np.empty(a.shape[0], np.float32)
    lib().orc_boxoverlap(_p(a), a.shape[0], _p(b), _p(out))
    return out


def nms(scored_boxes, thr, return_index=False):
    sb = _f32(scored_boxes).reshape(-1, 5)
    n = sb.shape[0]
    keep = np.empty((max(n, 1), 5), np.float32)
    idx = np.empty(max(n, 1), np.int32)
    k = lib().orc_nms(_p(sb), n, C.c_float(thr), _p(keep), _p(idx, i32p))
    return (keep[:k].copy(), idx[:k].copy()) if return_index else keep[:k].copy()


def nms_dense(boxes, overlap):
    """utils.nms_dense (utils.lua:402-462): 'another version of nms that returns indexes instead of new boxes'.  Restated
    tensor op for tensor op in float32 (boxes is a FloatTensor in demo.lua:85): sort by score descending; area = (x2-x1+1)*(y2-y1+1);
    for every unsuppressed c in sorted order: pick I[c]; xx1 = clamp(x1, x1[c], inf) ...; w = clamp(xx2 + (-1)*xx1 + 1, 0, inf);
    inter = w*h; union = area + (-1)*inter + area[c]; suppressed += (inter / union > overlap).  Returns the picks 1-based, as the Lua.
    torch.sort is TH's quicksort (not stable) and TH is absent: the order among bit-equal scores is PARITY UNPINNED; restated as
    ascending index (a stable sort)."""
    b = _f32(boxes).reshape(-1, 5)
    n = b.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    I = np.argsort(-b[:, 4], kind="stable")
    bs = b[I]
    x1, y1, x2, y2 = bs[:, 0], bs[:, 1], bs[:, 2], bs[:, 3]
    one, zero = np.float32(1), np.float32(0)
    area = ((x2 - x1) + one) * ((y2 - y1) + one)
    clamp = lambda v, lo, hi: np.where(v < lo, lo, np.where(v > hi, hi, v)).astype(np.float32)   # THTensor_(clamp)
    suppressed = np.zeros(n, bool)
    pick = []
    with np.errstate(all="ignore"):
        for c in range(n):
            if suppressed[c]:
                continue
            pick.append(int(I[c]) + 1)
            xx1, yy1 = clamp(x1, x1[c], np.float32(np.inf)), clamp(y1, y1[c], np.float32(np.inf))
            xx2, yy2 = clamp(x2, zero, x2[c]), clamp(y2, zero, y2[c])
            w = clamp((xx2 + np.float32(-1) * xx1) + one, zero, np.float32(np.inf))
            h = clamp((yy2 + np.float32(-1) * yy1) + one, zero, np.float32(np.inf))
            inter = w * h
            union = (area + np.float32(-1) * inter) + area[c]
            suppressed |= (inter / union) > np.float32(overlap)
    return np.asarray(pick, np.int64)


def bbox_vote(nms_boxes, scored_boxes, thr):
    nb, sb = _f32(nms_boxes).reshape(-1, 5), _f32(scored_boxes).reshape(-1, 5)
    res = np.zeros_like(nb)
    lib().orc_bbox_vote(_p(nb), nb.shape[0], _p(sb), sb.shape[0], C.c_float(thr), _p(res))
    return res


ROSS = dict(mean=(102.9801, 115.9465, 122.7717), std=None, scale=255.0, swap=(2, 1, 0))  # model_utils.lua:138-140
INCEPTION = dict(mean=(1.0, 1.0, 1.0), std=None, scale=2.0, swap=(0, 1, 2))  # fbcoco.ImageTransformer({1,1,1},nil,2), inceptionv3.lua:52
IMAGENET = dict(mean=(0.48462227599918, 0.45624044862054, 0.40588363755159),
                std=(0.22889466674951, 0.22446679341259, 0.22495548344775), scale=1.0, swap=(0, 1, 2))


def image_transform(im, mean, std=None, scale=1.0, swap=(0, 1, 2)):
    im = _f32(im)
    out = np.empty_like(im)
    sw = np.asarray(swap, np.int32)
    mn = np.asarray(mean, np.float64)
    sd = np.asarray(std if std is not None else (1, 1, 1), np.float64)
    lib().orc_image_transform(_p(im), im.shape[1], im.shape[2], _p(sw, i32p), C.c_double(scale), _p(mn, f64p),
                              _p(sd, f64p), int(std is not None), _p(out))
    return out


def image_scale(im, H2, W2):
    """image.scale(im, W2, H2) bilinear (unpinned restatement, see mpn_oracle.c)"""
    im = _f32(im)
    out = np.empty((im.shape[0], H2, W2), np.float32)
    lib().orc_image_scale(_p(im), im.shape[0], im.shape[1], im.shape[2], H2, W2, _p(out))
    return out


def pick_scale(H, W, target=600, max_size=1000):
    return float(lib().orc_pick_scale(H, W, C.c_double(target), C.c_double(max_size)))


def project_im_rois(boxes, scale):
    b = _f32(boxes).reshape(-1, 4)
    out = np.empty((b.shape[0], 5), np.float32)
    lib().orc_project_im_rois(_p(b), b.shape[0], C.c_double(scale), _p(out))
    return out


def conv3x3(x, w, b, relu=True):
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    Cin, H, W = x.shape
    Cout = w.shape[0]
    out = np.empty((Cout, H, W), np.float32)
    lib().orc_conv3x3(_p(x), Cin, H, W, _p(w), _p(b) if b is not None else None, Cout, int(relu), _p(out))
    return out


def maxpool2x2_ceil(x):
    x = _f32(x)
    Cc, H, W = x.shape
    out = np.empty((Cc, (H + 1) // 2, (W + 1) // 2), np.float32)
    lib().orc_maxpool2x2_ceil(_p(x), Cc, H, W, _p(out))
    return out


def linear(x, w, b, relu=False):
    x, w = _f32(x), _f32(w)
    b = _f32(b) if b is not None else None
    M, K = x.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    lib().orc_linear(_p(x), M, K, _p(w), _p(b) if b is not None else None, N, int(relu), _p(y))
    return y


def softmax(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_softmax(_p(x), x.shape[0], x.shape[1], _p(y))
    return y


ROI_BINS_CAFFE, ROI_BINS_ADAPTIVE = 0, 1
_roi_bin_rule = [ROI_BINS_CAFFE]   # what roi_pool() uses when its caller (the whole-model restatements below) does not say


class roi_bin_rule:
    """with O.roi_bin_rule(O.ROI_BINS_ADAPTIVE): ... — the whole-model restatements pool with inn.ROIPooling's CPU-branch rule inside."""

    def __init__(self, rule):
        self.rule = int(rule)

    def __enter__(self):
        _roi_bin_rule.append(self.rule)
        return self

    def __exit__(self, *a):
        _roi_bin_rule.pop()


def roi_pool(feat, rois, PH, PW, scale, coord_offset=1.0, end_adjust=0, bin_rule=None):
    """inn.ROIPooling; bin_rule = ROI_BINS_CAFFE (the CUDA branch, orc_roi_pool) | ROI_BINS_ADAPTIVE (the CPU branch: crop +
    SpatialAdaptiveMaxPooling, orc_roi_pool_rule)."""
    if bin_rule is None:
        bin_rule = _roi_bin_rule[-1]
    feat, rois = _f32(feat), _f32(rois).reshape(-1, 5)
    if feat.ndim == 3:
        feat = feat[None]
    B, Cc, H, W = feat.shape
    N = rois.shape[0]
    out = np.empty((N, Cc, PH, PW), np.float32)
    arg = np.empty((N, Cc, PH, PW), np.int32)
    lib().orc_roi_pool_rule(_p(feat), B, Cc, H, W, _p(rois), N, PH, PW, C.c_float(scale), C.c_float(coord_offset),
                            int(end_adjust), int(bin_rule), _p(out), _p(arg, i32p))
    return out, arg


def foveal(rois):
    r = _f32(rois).reshape(-1, 5)
    out = np.empty((4 * r.shape[0], 5), np.float32)
    lib().orc_foveal(_p(r), r.shape[0], _p(out))
    return out


def context_region(rois, scale):
    r = _f32(rois).reshape(-1, 5)
    out = np.empty_like(r)
    lib().orc_context_region(_p(r), r.shape[0], C.c_double(scale), _p(out))
    return out


def bbox_norm(bbox, mean4, std4):
    b = _f32(bbox).copy()
    m, s = _f32(mean4), _f32(std4)
    lib().orc_bbox_norm(_p(b), b.shape[0], b.shape[1], _p(m), _p(s))
    return b


def bbox_decode(boxes, deltas):
    bx, d = _f32(boxes).reshape(-1, 4), _f32(deltas)
    out = np.empty_like(d)
    lib().orc_bbox_decode(_p(bx), _p(d), bx.shape[0], d.shape[1] // 4, _p(out))
    return out


def clamp_boxes(bbox, im_w, im_h):
    b = _f32(bbox).copy()
    lib().orc_clamp_boxes(_p(b), C.c_size_t(b.size // 2), C.c_float(im_w), C.c_float(im_h))
    return b


def select_scored(scores, bbox, cls, thresh=-1.5):
    s, b = _f32(scores), _f32(bbox)
    n, Cc = s.shape
    sb = np.empty((max(n, 1), 5), np.float32)
    idx = np.empty(max(n, 1), np.int32)
    m = lib().orc_select_scored(_p(s), _p(b), n, Cc, cls, C.c_float(thresh), _p(sb), _p(idx, i32p))
    return sb[:m].copy(), idx[:m].copy()


def topk_threshold(scores, k):
    s = _f32(scores).ravel()
    return float(lib().orc_topk_threshold(_p(s), s.size, k))


def keep_top_k(per_class, k):
    """utils.lua:75-96 on a list of [K_j,5] arrays (one per class)."""
    allb = [b for b in per_class if b.size]
    if not allb:
        return per_class, 0.0
    t = topk_threshold(np.concatenate(allb)[:, 4], k)
    return [b[b[:, 4] >= t] if b.size else b for b in per_class], t


def select_boxes(scores, bbox):
    s, b = _f32(scores), _f32(bbox)
    out = np.empty((s.shape[0], 4), np.float32)
    lib().orc_select_boxes(_p(s), _p(b), s.shape[0], s.shape[1], _p(out))
    return out


def l2_normalize(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_l2_normalize(_p(x), x.shape[0], x.shape[1], _p(y))
    return y


def mean_over_k(probs):
    p = _f32(probs)
    out = np.empty(p.shape[1:], np.float32)
    lib().orc_mean_over_k(_p(p), p.shape[0], p.shape[1], p.shape[2], _p(out))
    return out


# ---------------------------------------------------------------- composed per-image path
# VGG-16 `features` (models/vgg.lua:14-27; layer indices multipathnet.lua:34-46): 13 conv3x3+ReLU,
# ceil-mode 2x2 pools after conv1_2, conv2_2, conv3_3, conv4_3; NO pool5.
VGG16_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512]


def vgg_trunk(x, conv_w, conv_b, cfg=None, taps=None):
    """x [3,H,W] -> conv5 map.  taps: optional dict collecting {'conv3','conv4','conv5'} outputs."""
    cfg = cfg or VGG16_CFG
    li = 0
    stage = 1
    for item in cfg:
        if item == "P":
            if taps is not None and stage in (3, 4):
                taps["conv%d" % stage] = x
            x = maxpool2x2_ceil(x)
            stage += 1
        else:
            x = conv3x3(x, conv_w[li], conv_b[li], relu=True)
            li += 1
    if taps is not None:
        taps["conv5"] = x
    return x


def frcnn_head(feat, rois, P, pooled=7, spatial_scale=1.0 / 16, chunk=None):
    """models/vgg.lua:28-31: ROIPooling -> View -> fc6/ReLU -> fc7/ReLU -> {cls, bbox}
    (+BBoxNorm, train.lua:136-138).  chunk=500 reproduces memoryEfficientForward (ImageDetect.lua:116-124)."""
    N = rois.shape[0]
    chunk = chunk or N
    cls, bbox = [], []
    for s in range(0, N, chunk):
        r = rois[s:s + chunk]
        pooledf, _ = roi_pool(feat, r, pooled, pooled, spatial_scale)
        x = pooledf.reshape(r.shape[0], -1)
        x = linear(x, P["fc6_w"], P["fc6_b"], relu=True)
        x = linear(x, P["fc7_w"], P["fc7_b"], relu=True)
        cls.append(linear(x, P["cls_w"], P["cls_b"]))
        bb = linear(x, P["bbox_w"], P["bbox_b"])
        if P.get("bbox_mean") is not None:
            bb = bbox_norm(bb, P["bbox_mean"], P["bbox_std"])
        bbox.append(bb)
    return np.concatenate(cls), np.concatenate(bbox)


def detect(im, boxes, P, transformer=ROSS, target=600, max_size=1000, cfg=None, pooled=7, chunk=500):
    """ImageDetect.lua:156-193.  im [3,H,W] fp32 in [0,1]; boxes [N,4] 1-based x1y1x2y2 in the ORIGINAL image.
    Returns (softmax scores [N,C], decoded boxes [N,4C], raw cls logits, raw deltas)."""
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    x = image_transform(im, **transformer)
    if s != 1.0:  # ImageDetect.lua:40-41: image.scale(im, W*s, H*s) on the transformed image
        x = image_scale(x, int(H * s), int(W * s))
    rois = project_im_rois(boxes, s)
    feat = vgg_trunk(x, P["conv_w"], P["conv_b"], cfg)
    logits, deltas = frcnn_head(feat, rois, P, pooled=pooled, chunk=chunk)
    dec = bbox_decode(boxes, deltas)
    return softmax(logits), dec, logits, deltas


def test_one(im, boxes, P, nms_thresh=0.3, score_thresh=-1.5, use_ref_nms=False, num_iter=1, use_rbox_scores=False,
             bbox_voting=False, bbox_vote_thresh=0.5, bbox_vote_score_pow=1.0, **kw):
    """Tester_FRCNN.lua:54-139: detect -> clamp the FIRST pass only (:75-78) -> for i = 2..num_iter: SelectBoxes on the previous
    pass, detect on the refined boxes (recompute_features = false: same trunk output; NOT clamped) (:82-89) ->
    opt.test_use_rbox_scores drops the first score table and the last box table (:91-97) -> joinTable -> per-class select ->
    NMS (-> bbox_vote, :118-124; the reference passes an unset threshold field there, we take bbox_vote_thresh; the votes are
    weighted by scores:pow(opt.test_bbox_voting_score_pow or 1) on a CLONE of the scored boxes (:119-121) — the kept boxes keep
    their NMS scores.  THFloatTensor_pow lives in the absent TH library (PARITY UNPINNED): restated as C `pow` on the float
    promoted to double, rounded back to float — what TH's generic `pow(*t_data, value)` does and, to the last bit except for
    one-in-2^29 double roundings, what its later x*x / sqrt special cases give for p = 2 / 0.5).
    Returns (list over classes 1..C-1 of [K,5]), (scores, boxes) as joined."""
    scores, dec, _, _ = detect(im, boxes, P, **kw)
    dec = clamp_boxes(dec, im.shape[2], im.shape[1])
    all_s, all_b = [scores], [dec]
    for _ in range(2, num_iter + 1):
        new_boxes = select_boxes(scores, dec)
        scores, dec, _, _ = detect(im, new_boxes, P, **kw)
        all_s.append(scores)
        all_b.append(dec)
    if use_rbox_scores:
        assert len(all_s) > 1
        all_s.pop(0)
        all_b.pop()
    scores, dec = np.concatenate(all_s), np.concatenate(all_b)
    out = []
    for j in range(1, scores.shape[1]):
        sb, _ = select_scored(scores, dec, j, score_thresh)
        kept = ref_nms(sb, nms_thresh) if use_ref_nms else nms(sb, nms_thresh)
        if bbox_voting:
            votes = sb
            if bbox_vote_score_pow != 1.0:
                votes = sb.copy()
                # THFloatTensor_pow(r, t, real value): the Lua double exponent arrives as a FLOAT
                votes[:, 4] = np.power(votes[:, 4].astype(np.float64), float(np.float32(bbox_vote_score_pow))).astype(np.float32)
            kept = (ref_bbox_vote if use_ref_nms else bbox_vote)(kept, votes, bbox_vote_thresh)
        out.append(kept)
    return out, (scores, dec)


# ---------------------------------------------------------------- MultiPathNet head (models/multipathnet.lua:64-120)
CONV345_NORM_FACTOR = (1.0, 1.0 / 30, 1.0 / 200)  # model_utils.lua:231-237: normFactor of conv5 / conv4 / conv3


def conv345_combine(maps, region_rois, T, pooled=7, spatial_scale=1.0 / 16, is_normalized=True):
    """model_utils.lua:209-251.  isNormalized=true: per map ROIPooling(7,7,scale_m) -> View(-1, C*49) -> nn.Normalize(2)
    -> View(-1,C,7,7); JoinTable(2) [conv5, conv4, conv3]; MulConstant(1000); 1x1 conv mix; View(-1).
    isNormalized=false (lines 222-223): nn.MulConstant(normFactor) per map instead of the normalisation, and no x1000
    (line 243 is inside `if isNormalized`).  MulConstant multiplies the FloatTensor by the Lua double: x * float32(factor) in fp32."""
    parts = []
    for m, use in enumerate((1, T["use4"], T["use3"])):
        if not use:
            continue
        pool, _ = roi_pool(maps[m], region_rois, pooled, pooled, spatial_scale * (2 ** m))
        n = pool.shape[0]
        if is_normalized:
            parts.append(l2_normalize(pool.reshape(n, -1)).reshape(pool.shape))
        else:
            parts.append(pool * np.float32(CONV345_NORM_FACTOR[m]))
    x = np.concatenate(parts, 1)                                          # [N, totalFeat, 7, 7]
    if is_normalized:
        x = x * np.float32(1000.0)
    n, tf = x.shape[:2]
    rows = np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(-1, tf))  # one row per (roi, bin): the 1x1 conv is a linear over channels
    y = linear(rows, T["mix_w"], T["mix_b"])                              # [N*49, c5]
    return np.ascontiguousarray(y.reshape(n, pooled * pooled, -1).transpose(0, 2, 1)).reshape(n, -1)  # flatten as [c5,7,7]


def mpnet_head(maps, rois, P, pooled=7, spatial_scale=1.0 / 16, return_raw=False):
    """maps = [conv5, conv4, conv3] ([C,h,w] each); rois [N,5].  Returns (scores [N,C] = mean of K softmaxes, bbox deltas [N,4C])
    (+ with return_raw the K classifiers' pre-softmax logits [N,K,C] and the towers' fc7 outputs side by side [N, towers * F])."""
    fov = foveal(rois).reshape(-1, 4, 5)
    outs = []
    for T in P["towers"]:
        x = conv345_combine(maps, np.ascontiguousarray(fov[:, T["region"]]), T, pooled, spatial_scale, is_normalized=P.get("conv345_norm", True))
        x = linear(x, T["fc6_w"], T["fc6_b"], relu=True)
        outs.append(linear(x, T["fc7_w"], T["fc7_b"], relu=True))
    cat = np.concatenate(outs[:-1], 1)                                    # ModelParallelTable concat along dim 2 + Narrow
    K, Cn = P["n_integral"], P["n_classes"]
    logits = linear(cat, P["cls_w"], P["cls_b"]).reshape(-1, K, Cn)
    probs = np.stack([softmax(np.ascontiguousarray(logits[:, k])) for k in range(K)])
    scores = mean_over_k(probs)
    deltas = linear(outs[-1], P["bbox_w"], P["bbox_b"])
    if P.get("bbox_mean") is not None:
        deltas = bbox_norm(deltas, P["bbox_mean"], P["bbox_std"])
    if return_raw:
        return scores, deltas, logits, np.concatenate(outs, 1)
    return scores, deltas


# ---- ResNet Fast R-CNN (models/resnet.lua; fb.resnet.torch topology, BN folded into the convolutions) -----------------
def bf16_round(x):
    """fp32 -> nearest-even bf16 -> fp32 (what the bf16 ResNet graph stores; finite inputs)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xffff0000)).view(np.float32)


def conv2d(x, w, b, stride=1, pad=0, relu=False, residual=None, bf16=False):
    """x [B,Cin,H,W], w [Cout,Cin,KH,KW] -> [B,Cout,OH,OW] (+ bias, + residual, ReLU).  bf16: weights rounded to bf16, fp32
    accumulation / bias / residual / ReLU, output rounded to bf16 (inputs are expected to be bf16-representable already)."""
    x, w = _f32(x), _f32(w)
    if bf16:
        w = bf16_round(w)
    b = _f32(b) if b is not None else None
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = np.empty((B, Cout, OH, OW), np.float32)
    res = _f32(residual) if residual is not None else None
    lib().orc_conv2d(_p(x), B, Cin, H, W, _p(w), _p(b) if b is not None else None, Cout, KH, KW, stride, pad,
                     _p(res) if res is not None else None, int(relu), _p(out))
    return bf16_round(out) if bf16 else out


def maxpool2d(x, k=3, stride=2, pad=1):
    x = _f32(x)
    B, Cc, H, W = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.empty((B, Cc, OH, OW), np.float32)
    lib().orc_maxpool2d(_p(x), B * Cc, H, W, k, stride, pad, _p(out))
    return out


def pool_out_size(h, k, stride, pad, ceil_mode):
    """nn.SpatialMaxPooling output size: floor mode, or :ceil() / Caffe (round up; the last window must start inside the padded input)"""
    if not ceil_mode:
        return (h + 2 * pad - k) // stride + 1
    o = -(-(h + 2 * pad - k) // stride) + 1
    if pad > 0 and (o - 1) * stride >= h + pad:
        o -= 1
    return o


def maxpool2d_mode(x, k, stride, pad, ceil_mode):
    x = _f32(x)
    B, Cc, H, W = x.shape
    OH, OW = pool_out_size(H, k, stride, pad, ceil_mode), pool_out_size(W, k, stride, pad, ceil_mode)
    out = np.empty((B, Cc, OH, OW), np.float32)
    lib().orc_maxpool2d_out(_p(x), B * Cc, H, W, k, stride, pad, OH, OW, _p(out))
    return out


def lrn(x, size=5, alpha=1e-4, beta=0.75, k=1.0):
    """nn.SpatialCrossMapLRN(size, alpha, beta, k) on [B,C,H,W]"""
    x = _f32(x)
    B, Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_lrn(_p(x), B, Cc, H * W, int(size), C.c_float(alpha), C.c_float(beta), C.c_float(k), _p(out))
    return out


def avgpool_global(x):
    x = _f32(x)
    B, Cc, H, W = x.shape
    out = np.empty((B, Cc), np.float32)
    lib().orc_avgpool_global(_p(x), B * Cc, H, W, _p(out))
    return out


def resnet_block(x, blk, bf16=False):
    """one residual block; blk = dict(convs=[(w,b,stride,pad), ...], shortcut=(w,b,stride) or None).  ReLU after every
    conv but the last; the last conv adds the shortcut, then ReLU (fb.resnet.torch basicblock / bottleneck)."""
    sc = x
    if blk["shortcut"] is not None:
        w, b, st = blk["shortcut"]
        sc = conv2d(x, w, b, stride=st, pad=0, relu=False, bf16=bf16)
    y = x
    n = len(blk["convs"])
    for i, (w, b, st, pd) in enumerate(blk["convs"]):
        last = i == n - 1
        y = conv2d(y, w, b, stride=st, pad=pd, relu=True, residual=sc if last else None, bf16=bf16)
    return y


def resnet_trunk(x, R):
    """net:get(1..7): conv1 7x7/2 (+BN folded) -> ReLU -> maxpool 3x3/2 pad 1 -> layer1..3.  x [3,H,W] -> [C3,H/16,W/16]"""
    bf = bool(R.get("bf16"))
    if bf:
        x = bf16_round(x)
    y = conv2d(x[None], R["conv1_w"], R["conv1_b"], stride=2, pad=3, relu=True, bf16=bf)
    y = maxpool2d(y, 3, 2, 1)
    for blk in R["trunk_blocks"]:
        y = resnet_block(y, blk, bf)
    return y[0]


def resnet_head(feat, rois, R, pooled=14, spatial_scale=1.0 / 16, chunk=None):
    """resnet.lua:40-48: ROIPooling(14,14,1/16) -> layer4 -> 7x7 average pool -> View -> {cls, bbox}"""
    N = rois.shape[0]
    chunk = chunk or N
    cls, bbox = [], []
    for s0 in range(0, N, chunk):
        r = rois[s0:s0 + chunk]
        y, _ = roi_pool(feat, r, pooled, pooled, spatial_scale)  # [n,C3,14,14]
        for blk in R["head_blocks"]:
            y = resnet_block(y, blk, bool(R.get("bf16")))
        f = avgpool_global(y)
        cls.append(linear(f, R["cls_w"], R["cls_b"]))
        bb = linear(f, R["bbox_w"], R["bbox_b"])
        if R.get("bbox_mean") is not None:
            bb = bbox_norm(bb, R["bbox_mean"], R["bbox_std"])
        bbox.append(bb)
    return np.concatenate(cls), np.concatenate(bbox)


def resnet_features(im, R, transformer=IMAGENET, target=600, max_size=1000):
    """the trunk's output for an image (what ImageDetect.lua:107-111 caches): pass it back as `feat=` to the *_detect functions"""
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    x = image_transform(im, **transformer)
    if s != 1.0:
        x = image_scale(x, int(H * s), int(W * s))
    return resnet_trunk(x, R)


def resnet_detect(im, boxes, R, transformer=IMAGENET, target=600, max_size=1000, pooled=14, chunk=None, feat=None):
    """ImageDetect.lua:156-193 on the ResNet model (ImagenetTransformer, resnet.lua:52)"""
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    rois = project_im_rois(boxes, s)
    if feat is None:
        feat = resnet_features(im, R, transformer, target, max_size)
    logits, deltas = resnet_head(feat, rois, R, pooled=pooled, chunk=chunk)
    dec = bbox_decode(boxes, deltas)
    return softmax(logits), dec, logits, deltas


def resnet_mpn_detect(im, boxes, R, transformer=IMAGENET, target=600, max_size=1000, pooled=14, return_raw=False, feat=None):
    """MultiPathNet on a ResNet backbone (this library's extension of multipathnet.lua:64-120 to resnet.lua's graph):
    Foveal regions -> per-tower ROIPooling + layer4 copy + average pool -> concat of the classification towers ->
    K classifier clones, mean of softmaxes; the last tower feeds the box regressor.  Returns (scores, decoded boxes)."""
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    rois = project_im_rois(boxes, s)
    if feat is None:
        feat = resnet_features(im, R, transformer, target, max_size)
    fov = foveal(rois).reshape(-1, 4, 5)
    outs = []
    for tw, rg in zip(R["head_towers"], R["head_regions"]):
        y, _ = roi_pool(feat, np.ascontiguousarray(fov[:, rg]), pooled, pooled, 1.0 / 16)
        for blk in tw:
            y = resnet_block(y, blk, bool(R.get("bf16")))
        outs.append(avgpool_global(y))
    cat = np.concatenate(outs[:-1], 1)
    K, Cn = R["n_integral"], R["n_classes"]
    logits = linear(cat, R["cls_w"], R["cls_b"]).reshape(-1, K, Cn)
    probs = np.stack([softmax(np.ascontiguousarray(logits[:, k])) for k in range(K)])
    scores = mean_over_k(probs)
    deltas = linear(outs[-1], R["bbox_w"], R["bbox_b"])
    if R.get("bbox_mean") is not None:
        deltas = bbox_norm(deltas, R["bbox_mean"], R["bbox_std"])
    if return_raw:  # + the K classifiers' pre-softmax logits [N, K, C], the pre-decode deltas, the towers' features [N, towers * F]
        return scores, bbox_decode(boxes, deltas), logits, deltas, np.concatenate(outs, 1)
    return scores, bbox_decode(boxes, deltas)


# ---- op-list graphs (Inception-v3, models/inceptionv3.lua:27-43) ---------------------------------------------------------------
def avgpool2d(x, k=3, stride=1, pad=1):
    """nn.SpatialAveragePooling, count_include_pad (always / k*k)"""
    x = _f32(x)
    B, Cc, H, W = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.empty((B, Cc, OH, OW), np.float32)
    lib().orc_avgpool2d(_p(x), B * Cc, H, W, k, stride, pad, _p(out))
    return out


def conv2d_rect(x, w, b, sh, sw, ph, pw, relu, bf16=False):
    x, w = _f32(x), _f32(w)
    if bf16:
        w = bf16_round(w)
    b = _f32(b) if b is not None else None
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    OH, OW = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    out = np.empty((B, Cout, OH, OW), np.float32)
    lib().orc_conv2d_rect(_p(x), B, Cin, H, W, _p(w), _p(b) if b is not None else None, Cout, KH, KW, sh, sw, ph, pw, int(relu), _p(out))
    return bf16_round(out) if bf16 else out


def graph_run(x0, ops, tensor_c, bf16=False):
    """executes an op list (include/mpn.h mpn_graph_op) on x0 [B,C0,H,W]; returns the list of tensors"""
    ts = [None] * len(tensor_c)
    ts[0] = x0
    for o in ops:
        x = ts[o["src"]]
        c0 = o.get("src_off", 0)
        if c0 or o["cin"] != x.shape[1]:  # a channel range of the source: one group of a grouped convolution
            x = np.ascontiguousarray(x[:, c0:c0 + o["cin"]])
        if o["kind"] == 0:
            y = conv2d_rect(x, o["w"], o["b"], o["sh"], o["sw"], o["ph"], o["pw"], o["relu"], bf16)
        elif o["kind"] == 1:
            y = maxpool2d_mode(x, o["kh"], o["sh"], o["ph"], o.get("ceil", 0))
        elif o["kind"] == 3:
            y = lrn(x, o["kh"], o["lrn"][0], o["lrn"][1], o["lrn"][2])
        else:
            y = avgpool2d(x, o["kh"], o["sh"], o["ph"])
            if bf16:
                y = bf16_round(y)
        d = o["dst"]
        if ts[d] is None:
            ts[d] = np.zeros((y.shape[0], tensor_c[d]) + y.shape[2:], np.float32)
        ts[d][:, o["off"]:o["off"] + y.shape[1]] = y
    return ts


def graph_features(im, G, transformer, target=600, max_size=1000):
    """the op-list trunk's output for an image (pass it back as `feat=` to graph_detect / graph_mpn_detect)"""
    bf = bool(G.get("bf16"))
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    x = image_transform(im, **transformer)
    if s != 1.0:
        x = image_scale(x, int(H * s), int(W * s))
    if bf:
        x = bf16_round(x)
    return graph_run(x[None], G["trunk_ops"], G["trunk_tensor_c"], bf)[G["feat_tensor"]][0]


def graph_detect(im, boxes, G, transformer, target=600, max_size=1000, pooled=17, spatial_scale=17.0 / 299.0, feat=None):
    """ImageDetect.lua:156-193 on an op-list model (Inception-v3 Fast R-CNN)"""
    bf = bool(G.get("bf16"))
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    rois = project_im_rois(boxes, s)
    if feat is None:
        feat = graph_features(im, G, transformer, target, max_size)
    pooledf, _ = roi_pool(feat, rois, pooled, pooled, spatial_scale)
    y = graph_run(pooledf, G["head_ops"], G["head_tensor_c"], bf)[G["out_tensor"]]
    f = avgpool_global(y)
    logits = linear(f, G["cls_w"], G["cls_b"])
    deltas = linear(f, G["bbox_w"], G["bbox_b"])
    if G.get("bbox_mean") is not None:
        deltas = bbox_norm(deltas, G["bbox_mean"], G["bbox_std"])
    return softmax(logits), bbox_decode(boxes, deltas), logits, deltas


def graph_mpn_detect(im, boxes, G, transformer, target=600, max_size=1000, pooled=17, spatial_scale=17.0 / 299.0, return_raw=False, feat=None):
    """MultiPathNet towers on an op-list backbone (this library's extension, cf. resnet_mpn_detect): returns (scores, decoded boxes)"""
    bf = bool(G.get("bf16"))
    H, W = im.shape[1:]
    s = pick_scale(H, W, target, max_size)
    rois = project_im_rois(boxes, s)
    if feat is None:
        feat = graph_features(im, G, transformer, target, max_size)
    fov = foveal(rois).reshape(-1, 4, 5)
    outs = []
    for tw, rg in zip(G["head_towers"], G["head_regions"]):
        pooledf, _ = roi_pool(feat, np.ascontiguousarray(fov[:, rg]), pooled, pooled, spatial_scale)
        outs.append(avgpool_global(graph_run(pooledf, tw, G["head_tensor_c"], bf)[G["out_tensor"]]))
    cat = np.concatenate(outs[:-1], 1)
    K, Cn = G["n_integral"], G["n_classes"]
    logits = linear(cat, G["cls_w"], G["cls_b"]).reshape(-1, K, Cn)
    probs = np.stack([softmax(np.ascontiguousarray(logits[:, k])) for k in range(K)])
    deltas = linear(outs[-1], G["bbox_w"], G["bbox_b"])
    if G.get("bbox_mean") is not None:
        deltas = bbox_norm(deltas, G["bbox_mean"], G["bbox_std"])
    if return_raw:
        return mean_over_k(probs), bbox_decode(boxes, deltas), logits, deltas, np.concatenate(outs, 1)
    return mean_over_k(probs), bbox_decode(boxes, deltas)


# ---- wire formats either side of the path ----------------------------------------------------------------------
def filter_area(boxes, scores, area):
    """DataSetJSON.lua:171-186 filterArea: boxes are {y1,x1,y2,x2} rows; keep rows with (col3-col1)*(col4-col2) > area, in row
    order (`s:gt(area):nonzero()`); area == 0 returns everything (line 172)."""
    boxes = np.asarray(boxes, np.float32)
    if area == 0:
        return boxes, scores
    wh = boxes[:, 2:4] - boxes[:, 0:2]                       # narrow(2,3,2):clone():add(-1, narrow(2,1,2)), fp32
    idx = np.nonzero(wh[:, 0] * wh[:, 1] > np.float32(area))[0]
    return boxes[idx], (None if scores is None else np.asarray(scores, np.float32)[idx])


def filter_score(boxes, scores, best_number):
    """DataSetJSON.lua:157-169 filterScore: when there are more than best_number rows, `scores:sort(true)` and the first
    best_number indices.  Torch7's sort is TH's quicksort, whose order AMONG EQUAL scores is an implementation detail of a
    library that is not in the reference tree (unpinned); this restatement — and the device path — break ties by the lower
    original index (a stable sort).  With distinct scores the order is fully determined."""
    if scores is None:
        return boxes, None
    boxes, scores = np.asarray(boxes, np.float32), np.asarray(scores, np.float32)
    if best_number is not None and boxes.shape[0] > best_number:
        idx = np.argsort(-scores.astype(np.float64), kind="stable")[:best_number]
        boxes, scores = boxes[idx], scores[idx]
    return boxes, scores


def prepare_proposals(boxes_yxyx, scores=None, min_area=0.0, best_number=None):
    """DataSetJSON.lua:216-233 (loadROIDB's per-image body): float() -> filterArea -> filterScore -> index(2, {2,1,4,3})."""
    b, s = filter_area(boxes_yxyx, scores, min_area)
    b, s = filter_score(b, s, best_number)
    b = np.asarray(b, np.float32).reshape(-1, 4)
    return b[:, [1, 0, 3, 2]], s


def coco_rows(dets, image_id, category_ids):
    """testCoco/init.lua:69-85: per detection {x1,y1,x2,y2,score,class(1-based)} -> {image id, x1-1, y1-1, x2-x1, y2-y1, score,
    categories.id[class]} (fp32 arithmetic, as the FloatTensor `boxt` holds it)."""
    d = np.asarray(dets, np.float32).reshape(-1, 6)
    cats = np.asarray(category_ids, np.float32)
    one = np.float32(1.0)
    return np.stack([np.full(d.shape[0], np.float32(image_id), np.float32), d[:, 0] - one, d[:, 1] - one, d[:, 2] - d[:, 0], d[:, 3] - d[:, 1],
                     d[:, 4], cats[d[:, 5].astype(np.int64) - 1]], 1).astype(np.float32)


def save_results_table(aboxes, dataset_name):
    """utils.lua:335-372 saveResults: aboxes[class][image] = [K,5] -> flat boxes / scores / categories / images tables, class-major,
    images inside a class in index order, empty entries skipped."""
    boxes, scores, cats, imgs = [], [], [], []
    for cls, per_img in enumerate(aboxes, start=1):
        for i, data in enumerate(per_img, start=1):
            if data is not None and np.asarray(data).size > 0:
                data = np.asarray(data, np.float32).reshape(-1, 5)
                boxes.append(data[:, :4]); scores.append(data[:, 4])
                cats.append(np.full(data.shape[0], cls, np.float32)); imgs.append(np.full(data.shape[0], i, np.float32))
    cat = lambda xs, w: np.concatenate(xs) if xs else np.zeros((0,) + w, np.float32)
    n_images = len(aboxes[0]) if aboxes else 0
    return {"dataset": dataset_name, "images": np.arange(1, n_images + 1, dtype=np.float32),
            "detections": {"boxes": cat(boxes, (4,)), "scores": cat(scores, ()), "categories": cat(cats, ()), "images": cat(imgs, ())}}
