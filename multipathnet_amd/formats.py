"""Data formats either side of the hot path, mirroring the reference's Lua helpers:

  prepare_proposals     DataSetJSON.lua:157-239  filterArea -> filterScore (best_n by score) -> {2,1,4,3} column permute
  detections_to_coco    testCoco/init.lua:65-85  per-class [K,5] tables -> [n,7] rows {img, x, y, w, h, score, category}
  save_results          utils.lua:335-372        flat boxes / scores / categories / images tables
  load_proposals        DataSetJSON.lua:124-160  torch.load of one or several proposal `.t7` files, merged by image name
  roidb_from_proposals  DataSetJSON.lua:188-239  per dataset image: float() -> filterArea -> filterScore -> permute
  save_boxes / load_boxes, save_results_file    run_test.lua:62,72-76 (`boxes.t7`), utils.lua:335-372 (torch.save of the flat table)

The `.t7` files are read and written by multipathnet_amd/t7.py (Torch7's binary serialisation), so real proposal tables flow in
and `boxes.t7` / results tables flow out to the reference's evaluation scripts unchanged.

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
        _, order = sc.sort(descending=True, stable=True)  # equal scores: lower row first (Torch7's own tie order is unspecified)
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


def load_proposals(roidbfile):
    """DataSetCOCO:loadAndMergeProposals (DataSetJSON.lua:124-160): one `.t7` path, or a list of them merged by image name
    (boxes / scores of an image concatenated in file order; files without scores count as score 0).
    Returns {"boxes": [ndarray [n,4] per image], "scores": [...] or None, "images": [str]}."""
    import numpy as np
    from . import t7
    if isinstance(roidbfile, str):
        dt = t7.load(roidbfile)
        return {"boxes": list(dt["boxes"]), "scores": list(dt["scores"]) if dt.get("scores") is not None else None, "images": list(dt["images"])}
    out = {"boxes": [], "scores": [], "images": []}
    img2idx = {}
    for path in roidbfile:
        dt2 = t7.load(path)
        for k, v in enumerate(dt2["images"]):
            if v not in img2idx:
                out["images"].append(v)
                out["boxes"].append(None)
                out["scores"].append(None)
                img2idx[v] = len(out["images"]) - 1
            i = img2idx[v]
            b = np.asarray(dt2["boxes"][k], np.float32)
            sc = np.asarray(dt2["scores"][k], np.float32).reshape(-1) if dt2.get("scores") is not None else np.zeros(b.shape[0], np.float32)
            out["boxes"][i] = _table_concat(out["boxes"][i], b)
            out["scores"][i] = _table_concat(out["scores"][i], sc)
    return out


def _table_concat(t1, t2):
    """TableConcat (DataSetJSON.lua:114-122): an absent or EMPTY operand (an image without proposals in one of the merged files
    is a 0-element tensor of no particular shape) yields the other one as float; otherwise torch.cat along dim 1."""
    import numpy as np
    if t1 is None or t1.size == 0:
        return np.asarray(t2, np.float32)
    if t2 is None or t2.size == 0:
        return np.asarray(t1, np.float32)
    return np.concatenate([np.asarray(t1, np.float32), np.asarray(t2, np.float32)], 0)


def roidb_from_proposals(dt, file_names, best_number=None, min_area=0.0, allow_missing=False, device=None):
    """DataSetCOCO:loadROIDB (DataSetJSON.lua:188-239): for every dataset image (by file name) its proposals as float {x1,y1,x2,y2}
    boxes after filterArea / filterScore / the {2,1,4,3} permute, on `device` (rows through mpn_proposals_permute_filter).
    Returns (roidb list, scoredb list); a missing image raises unless allow_missing (then None)."""
    im2box = {name: i for i, name in enumerate(dt["images"])}
    if dt.get("scores") is not None:
        assert len(dt["boxes"]) == len(dt["scores"])
        assert isinstance(best_number, (int, float)), "best_number has to be a valid number, e.g. 500 or 5000"
    dev = device or torch.device("cuda", torch.cuda.current_device())
    roidb, scoredb = [], []
    for name in file_names:
        if name not in im2box:
            if not allow_missing:
                raise KeyError(name + " is not in proposals")
            roidb.append(None)
            scoredb.append(None)
            continue
        k = im2box[name]
        b = torch.as_tensor(dt["boxes"][k], dtype=torch.float32).to(dev)
        sc = torch.as_tensor(dt["scores"][k], dtype=torch.float32).reshape(-1).to(dev) if dt.get("scores") is not None else None
        boxes, scores = prepare_proposals(b, sc, min_area=min_area, best_number=int(best_number) if best_number is not None else None)
        roidb.append(boxes)
        scoredb.append(scores)
    return roidb, scoredb


def save_boxes(path, aboxes):
    """run_test.lua:72-76 `torch.save(dir/boxes.t7, aboxes)`: aboxes[class][image] = FloatTensor [K,5] (empty = no detections)."""
    import numpy as np
    from . import t7
    t7.save(path, [[(np.zeros((0,), np.float32) if (d is None or len(d) == 0) else np.asarray(d.detach().cpu() if hasattr(d, "detach") else d, np.float32))
                    for d in per_img] for per_img in aboxes])


def load_boxes(path):
    """run_test.lua:62 `aboxes = torch.load(opt.test_load_aboxes)` -> aboxes[class][image] as float32 arrays ([0,5] when empty)."""
    import numpy as np
    from . import t7
    ab = t7.load(path)
    fix = lambda d: np.zeros((0, 5), np.float32) if d is None or np.size(d) == 0 else np.asarray(d, np.float32)
    cls_tables = ab if isinstance(ab, list) else [ab[k] for k in sorted(ab)]
    return [[fix(d) for d in (per if isinstance(per, list) else [per[k] for k in sorted(per)])] for per in cls_tables]


def save_results_file(path, aboxes, dataset_name):
    """utils.saveResults (utils.lua:335-372): the flat table, torch.save()d."""
    from . import t7
    t7.save(path, save_results(aboxes, dataset_name))
