"""Host-side mirror of the evaluation helpers of utils.lua (file:line relative to /root/reference).

  utils.nms / utils.bbox_vote   utils.lua:29-39  (FFI to nms.c -> here: wavefront kernels)
  utils.nms_dense               utils.lua:402-462 (index-returning NMS of demo.lua)
  utils.boxoverlap              utils.lua:104-128 (same formula as nms.c:14-41)
  utils.convertFrom             utils.lua:212-248
  utils.keep_top_k              utils.lua:75-96
  utils.joinTable               utils.lua:46-71
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .nn import _f, _i, _stream


def nms(boxes, overlap):
    """utils.nms(boxes [M,5] {x1,y1,x2,y2,score}, overlap) -> kept rows in selection order."""
    assert boxes.dim() == 2 and boxes.size(1) == 5 if boxes.numel() else True
    M = boxes.size(0) if boxes.numel() else 0
    if M == 0:
        return torch.empty((0, 5), dtype=torch.float32, device=boxes.device)
    keep = torch.empty((M, 5), dtype=torch.float32, device=boxes.device)
    n = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    check(_lib.load().mpn_nms(_f(boxes, "boxes"), M, C.c_float(overlap), _f(keep), None, _i(n), _stream()), "nms")
    return keep[: int(n.item())]


def nms_with_index(boxes, overlap):
    M = boxes.size(0)
    keep = torch.empty((max(M, 1), 5), dtype=torch.float32, device=boxes.device)
    idx = torch.empty(max(M, 1), dtype=torch.int32, device=boxes.device)
    n = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    if M:
        check(_lib.load().mpn_nms(_f(boxes), M, C.c_float(overlap), _f(keep), _i(idx), _i(n), _stream()), "nms")
    k = int(n.item())
    return keep[:k], idx[:k]


def nms_dense(boxes, overlap):
    """utils.nms_dense(boxes [M,5], overlap) (utils.lua:402-462) -> LongTensor of picked row indices, 1-BASED like the Lua
    function's (subtract 1 to index a torch tensor), in pick order.  Any table width (beyond 8192 rows: a counting-rank + sequential-walk
    form).  Order among bit-equal scores: ascending index; NaN scores sort last (TH's quicksort order is unpinned, TH absent)."""
    M = boxes.size(0) if boxes.numel() else 0
    if M == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    assert boxes.dim() == 2 and boxes.size(1) == 5   # utils.lua:411
    pick = torch.empty(M, dtype=torch.int32, device=boxes.device)
    n = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    check(_lib.load().mpn_nms_dense(_f(boxes, "boxes"), M, C.c_float(overlap), _i(pick), _i(n), _stream()), "nms_dense")
    return pick[: int(n.item())].to(torch.int64)


def nms_batched(scored, counts, overlap):
    """scored [n_cls, M, 5], counts [n_cls] int32 (or None) -> (keep [n_cls,M,5], keep_idx, n_keep)."""
    n_cls, M, _ = scored.shape
    keep = torch.empty_like(scored)
    idx = torch.empty((n_cls, M), dtype=torch.int32, device=scored.device)
    n = torch.zeros(n_cls, dtype=torch.int32, device=scored.device)
    check(_lib.load().mpn_nms_batched(_f(scored), _i(counts) if counts is not None else None, n_cls, M, C.c_float(overlap),
                                      _f(keep), _i(idx), _i(n), _stream()), "nms_batched")
    return keep, idx, n


def bbox_vote(nms_boxes, scored_boxes, overlap):
    """utils.bbox_vote(nms_boxes [K,5], scored_boxes [M,5], threshold) -> [K,5]."""
    K = nms_boxes.size(0)
    res = torch.zeros((K, 5), dtype=torch.float32, device=nms_boxes.device)
    if K:
        assert nms_boxes.is_contiguous() and scored_boxes.is_contiguous()  # nms.c:112-113
        assert nms_boxes.size(1) == 5 and scored_boxes.size(1) == 5        # nms.c:119-120
        check(_lib.load().mpn_bbox_vote(_f(nms_boxes), K, None, _f(scored_boxes), scored_boxes.size(0), C.c_float(overlap),
                                        _f(res), _stream()), "bbox_vote")
    return res


def boxoverlap(a, b):
    """utils.boxoverlap(a [N,4], b {x1,y1,x2,y2}) -> IoU [N] (the +1 pixel convention)."""
    bb = (C.c_float * 4)(*[float(v) for v in b])
    out = torch.empty(a.size(0), dtype=torch.float32, device=a.device)
    if a.size(0):
        check(_lib.load().mpn_boxoverlap(_f(a.contiguous()), a.size(0), bb, _f(out), _stream()), "boxoverlap")
    return out


def convertFrom(out, bbox, y):
    """utils.convertFrom(out, bbox [N,4], y [N,4]) — decodes regression deltas into `out` (may alias y)."""
    assert bbox.size(1) == y.size(1) and bbox.size(0) == y.size(0)
    res = torch.empty_like(y)
    if y.size(0):
        check(_lib.load().mpn_bbox_decode(_f(bbox.contiguous()), _f(y.contiguous()), y.size(0), 1, _f(res), _stream()),
              "convertFrom")
    out.copy_(res)
    return out


def decode_all_classes(boxes, deltas):
    """ImageDetect.lua:183-185: convertFrom over every 4-column class block at once."""
    res = torch.empty_like(deltas)
    if deltas.size(0):
        check(_lib.load().mpn_bbox_decode(_f(boxes), _f(deltas), deltas.size(0), deltas.size(1) // 4, _f(res), _stream()),
              "convertFrom")
    return res


def joinTable(input, dim=0):
    xs = [t for t in input if t.numel() > 0]
    return torch.cat(xs, dim) if xs else input[0].new_empty((0,))


def keep_top_k(boxes, top_k):
    """utils.keep_top_k(boxes = list of [K_j,5] per class, k) -> (filtered list, thresh)."""
    X = [b for b in boxes if b.numel() > 0]
    if not X:
        return boxes, 0
    n_cls = len(boxes)
    M = max(b.size(0) for b in X)
    dev = X[0].device
    keep = torch.zeros((n_cls, M, 5), dtype=torch.float32, device=dev)
    n = torch.zeros(n_cls, dtype=torch.int32, device=dev)
    for j, b in enumerate(boxes):
        if b.numel():
            keep[j, : b.size(0)] = b
            n[j] = b.size(0)
    total = int(n.sum().item())
    out = torch.empty((total, 6), dtype=torch.float32, device=dev)
    thr = torch.zeros(1, dtype=torch.float32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    check(_lib.load().mpn_keep_top_k(_f(keep), _i(n), n_cls, M, int(top_k), _f(thr), _f(out), total, _i(n_out), _stream()),
          "keep_top_k")
    out = out[: int(n_out.item())]
    res = []
    for j in range(n_cls):
        sel = out[out[:, 5] == float(j + 1)]
        res.append(sel[:, :5].contiguous() if sel.numel() else boxes[j].new_empty((0, 5)))
    return res, float(thr.item())
