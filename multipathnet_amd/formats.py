"""Data formats either side of the hot path, mirroring the reference's Lua helpers:

  prepare_proposals     DataSetJSON.lua:157-239  filterArea -> filterScore (best_n by score) -> {2,1,4,3} column permute
  detections_to_coco    testCoco/init.lua:65-85  per-class [K,5] tables -> [n,7] rows {img, x, y, w, h, score, category}
  save_results          utils.lua:335-372        flat boxes / scores / categories / images tables

Host orchestration is torch indexing; the per-row arithmetic runs in libmpn_hip.so (mpn_proposals_permute_filter,
mpn_dets_to_coco_rows).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .nn import _f, _i, _stream


def prepare_proposals(boxes_yxyx, scores=None, min_area=0.0, best_number=None):
    """boxes_yxyx [N,4] as stored in the proposal tables ({y1,x1,y2,x2}); returns (boxes [M,4] {x1,y1,x2,y2}, scores [M] or None)."""
    b = boxes_yxyx.to(torch.float32).contiguous()
    n = b.size(0)
    if b.dim() != 2 or b.size(1) != 4:
        return torch.empty((0, 4), dtype=torch.float32, device=b.device), scores
    out = torch.empty_like(b)
    keep = torch.ones(max(n, 1), dtype=torch.int32, device=b.device)
    if n:
        check(_lib.load().mpn_proposals_permute_filter(_f(b), n, C.c_float(min_area), _f(out), _i(keep), _stream()), "prepare_proposals")
    idx = keep[:n].nonzero().view(-1)            # filterArea keeps row order
    out = out.index_select(0, idx)
    sc = scores.to(torch.float32).reshape(-1).index_select(0, idx) if scores is not None else None
    if sc is not None and best_number is not None and out.size(0) > best_number:  # filterScore
        _, order = sc.sort(descending=True)
        order = order[:best_number]
        out, sc = out.index_select(0, order), sc.index_select(0, order)
    return out.contiguous(), sc


def detections_to_coco(dets, n_dets, image_id, category_ids=None):
    """dets [cap,6] {x1,y1,x2,y2,score,class} + device count -> [n,7] COCO rows (0-based x,y; w,h without +1)."""
    cap = dets.size(0)
    rows = torch.zeros((cap, 7), dtype=torch.float32, device=dets.device)
    cat = torch.tensor(category_ids, dtype=torch.float32, device=dets.device) if category_ids is not None else None
    check(_lib.load().mpn_dets_to_coco_rows(_f(dets), _i(n_dets) if n_dets is not None else None, cap, C.c_float(image_id),
                                            _f(cat) if cat is not None else None, len(category_ids) if category_ids is not None else 0,
                                            _f(rows), _stream()), "detections_to_coco")
    n = int(n_dets.item()) if n_dets is not None else cap
    return rows[: min(n, cap)]


def save_results(aboxes, dataset_name):
    """utils.saveResults layout: aboxes[class][image] = [K,5]; returns the table the reference torch.save()s."""
    boxes, scores, cats, imgs = [], [], [], []
    for cls, per_img in enumerate(aboxes, start=1):
        for i, data in enumerate(per_img, start=1):
            if data is not None and data.numel() > 0:
                boxes.append(data[:, :4])
                scores.append(data[:, 4])
                cats.append(torch.full((data.size(0),), float(cls)))
                imgs.append(torch.full((data.size(0),), float(i)))
    n_images = len(aboxes[0]) if aboxes else 0
    cat = lambda xs, w: torch.cat([x.cpu() for x in xs]) if xs else torch.empty((0,) + w)
    return {"dataset": dataset_name, "images": torch.arange(1, n_images + 1, dtype=torch.float32),
            "detections": {"boxes": cat(boxes, (4,)), "scores": cat(scores, ()), "categories": cat(cats, ()), "images": cat(imgs, ())}}
