"""fbcoco.ImageDetect and fbcoco.Tester_FRCNN mirrors (ImageDetect.lua, Tester_FRCNN.lua).

`ImageDetect.detect(im, boxes)` returns what ImageDetect.lua:156-193 returns — softmax class
scores [N,C] and decoded boxes [N,4C] — and `Tester_FRCNN.testOne` what Tester_FRCNN.lua:54-139
returns: per-class NMS'd boxes plus the raw {output, bbox_pred}.  All arithmetic runs in the fused
device pipeline (models.FastRCNN); this file is orchestration only.
"""
import time

import torch

from . import _lib, utils
from .nn import SelectBoxes


class ImageDetect(object):
    def __init__(self, model, transformer=None, scale=None, max_size=None):
        assert model is not None, "must provide model!"
        self.model = model
        self.image_transformer = transformer  # the pipeline applies it on the device (cfg.tf_*)
        self.scale = scale or [600]
        self.max_size = max_size or 1000

    def detect(self, im, boxes, min_images=None, recompute_features=True):
        """ImageDetect.lua:156-193.  Rescaling (getImages), ROI projection, decode on the original boxes all run inside the
        device pipeline; the model must have been built with the same scale / max_size."""
        im = im.to(self.model.device, torch.float32).contiguous()
        boxes = boxes.to(self.model.device, torch.float32).contiguous()
        return self.model.detect(im, boxes, recompute_features=recompute_features)


class Tester_FRCNN(object):
    def __init__(self, module, transformer=None, dataset=None, scale=None, max_size=None, opt=None):
        opt = opt or {}
        self.dataset = dataset
        self.module = module
        self.detec = ImageDetect(module, transformer, scale, max_size)
        self.num_iter = opt.get("test_num_iterative_loc", 1)
        self.nms_thresh = opt.get("test_nms_threshold", 0.3)
        self.bbox_vote_thresh = opt.get("test_bbox_voting_nms_threshold", 0.5)
        self.test_bbox_voting = opt.get("test_bbox_voting", False)
        self.test_bbox_voting_score_pow = opt.get("test_bbox_voting_score_pow", 1)
        self.num_classes = module.n_classes - 1
        self.thresh = -1.5  # Tester_FRCNN.lua:50
        self.boxselect = SelectBoxes()
        self.last_timing = {}

    def testOne(self, im, boxes):
        """im [3,H,W], boxes [N,4] -> (list over classes of [K,5], (output, bbox_pred))."""
        t0 = time.time()
        output, bbox_pred = self.detec.detect(im, boxes)
        all_output, all_bbox = [output], [bbox_pred]
        for _ in range(2, self.num_iter + 1):  # Tester_FRCNN.lua:82-89 iterative localisation
            new_boxes = self.boxselect.forward([output, bbox_pred])
            output, bbox_pred = self.detec.detect(im, new_boxes, None, False)  # recompute_features = false (:87)
            all_output.append(output)
            all_bbox.append(bbox_pred)
        output = utils.joinTable(all_output, 0)
        bbox_pred = utils.joinTable(all_bbox, 0)
        torch.cuda.synchronize()
        t1 = time.time()
        # Tester_FRCNN.lua:106-125, all classes in one launch
        import ctypes as C
        from .nn import _f, _i, _stream
        N, Cc = output.shape
        scored = torch.empty((Cc - 1, N, 5), dtype=torch.float32, device=output.device)
        counts = torch.zeros(Cc - 1, dtype=torch.int32, device=output.device)
        _lib.check(_lib.load().mpn_select_scored(_f(output), _f(bbox_pred), N, Cc, 1, C.c_float(self.thresh), _f(scored),
                                                 _i(counts), None, _stream()), "select_scored")
        keep, _, n_keep = utils.nms_batched(scored, counts, self.nms_thresh)
        nk = n_keep.tolist()
        cnt = counts.tolist()
        img_boxes = []
        for j in range(Cc - 1):
            kb = keep[j, : nk[j]]
            if self.test_bbox_voting:
                sb = scored[j, : cnt[j]].clone()
                sb[:, 4] = sb[:, 4] ** self.test_bbox_voting_score_pow
                kb = utils.bbox_vote(kb.contiguous(), sb.contiguous(), self.bbox_vote_thresh)
            img_boxes.append(kb)
        torch.cuda.synchronize()
        t2 = time.time()
        self.last_timing = {"forward": t1 - t0, "nms": t2 - t1, "total": t2 - t0}  # Tester_FRCNN.lua:130-136
        return img_boxes, (output, bbox_pred)

    def keepTopKPerImage(self, img_boxes, k=100):
        return utils.keep_top_k(img_boxes, k)[0]
