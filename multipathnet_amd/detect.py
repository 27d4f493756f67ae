"""fbcoco.ImageDetect and fbcoco.Tester_FRCNN mirrors (ImageDetect.lua, Tester_FRCNN.lua).

`ImageDetect.detect(im, boxes)` returns what ImageDetect.lua:156-193 returns — softmax class
scores [N,C] and decoded boxes [N,4C] — and `Tester_FRCNN.testOne` what Tester_FRCNN.lua:54-139
returns: per-class NMS'd boxes plus the raw {output, bbox_pred}.  All arithmetic runs in the fused
device pipeline (models.FastRCNN); this file is orchestration only.
"""
import ctypes as C
import time

import torch

from . import _lib, utils
from .nn import SelectBoxes, _f, _i, _stream


class ImageDetect(object):
    def __init__(self, model, transformer=None, scale=None, max_size=None):
        assert model is not None, "must provide model!"
        self.model = model
        self.image_transformer = transformer  # the pipeline applies it on the device (cfg.tf_*)
        self.scale = scale or [600]
        self.max_size = max_size or 1000
        # getImages' rescaling (ImageDetect.lua:34-43) runs inside the device pipeline with the MODEL's scale / max_size: an
        # ImageDetect asked for another configuration would silently skip or change the resize — refuse instead.
        m_scale, m_max = getattr(model, "scale", None), getattr(model, "max_size", None)
        if scale is not None or max_size is not None:
            want = (float(self.scale[0]), float(self.max_size))
            have = (float(m_scale), float(m_max or 0)) if m_scale else None
            if have != want:
                raise ValueError("ImageDetect(scale=%r, max_size=%r): the model's device pipeline was built with scale=%r, max_size=%r; "
                                 "build models.FastRCNN(..., scale=%r, max_size=%r)" % (scale, max_size, m_scale, m_max, self.scale[0], self.max_size))

    def detect(self, im, boxes, min_images=None, recompute_features=True):
        """ImageDetect.lua:156-193: (softmax scores [N,C], decoded boxes [N,4C]) — NOT clamped to the image, as in the
        reference (the clamp is Tester_FRCNN's, on its first detect() only).  Rescaling (getImages), ROI projection and the
        decode on the original boxes all run inside the device pipeline."""
        im = im.to(self.model.device, torch.float32).contiguous()
        boxes = boxes.to(self.model.device, torch.float32).contiguous()
        return self.model.detect(im, boxes, recompute_features=recompute_features, clamp=False)


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
        self.test_use_rbox_scores = opt.get("test_use_rbox_scores", False)
        self.num_classes = module.n_classes - 1
        self.thresh = -1.5  # Tester_FRCNN.lua:50
        self.boxselect = SelectBoxes()
        self.last_timing = {}

    def testOne(self, im, boxes):
        """im [3,H,W], boxes [N,4] -> (list over classes of [K,5], (output, bbox_pred))."""
        t0 = time.time()
        output, bbox_pred = self.detec.detect(im, boxes)
        # clamp predictions within image (Tester_FRCNN.lua:75-78): in place, the FIRST pass only
        _lib.check(_lib.load().mpn_clamp_boxes(_f(bbox_pred), C.c_size_t(bbox_pred.numel() // 2), C.c_float(im.shape[2]),
                                               C.c_float(im.shape[1]), _stream()), "clamp")
        all_output, all_bbox = [output], [bbox_pred]
        for _ in range(2, self.num_iter + 1):  # Tester_FRCNN.lua:82-89 iterative localisation
            new_boxes = self.boxselect.forward([output, bbox_pred])
            output, bbox_pred = self.detec.detect(im, new_boxes, None, False)  # recompute_features = false (:87)
            all_output.append(output)
            all_bbox.append(bbox_pred)
        if self.test_use_rbox_scores:  # Tester_FRCNN.lua:91-97: scores of pass i+1 for the boxes of pass i
            assert len(all_output) > 1
            all_output.pop(0)
            all_bbox.pop()
        output = utils.joinTable(all_output, 0)
        bbox_pred = utils.joinTable(all_bbox, 0)
        torch.cuda.synchronize()
        t1 = time.time()
        # Tester_FRCNN.lua:106-125, all classes in one launch
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
                if self.test_bbox_voting_score_pow != 1:  # scores:pow(p) as THFloatTensor_pow: C pow in double, rounded once
                    # (the exponent reaches THFloatTensor_pow as a float: `real value`)
                    p32 = float(torch.tensor(float(self.test_bbox_voting_score_pow), dtype=torch.float32))
                    sb[:, 4] = sb[:, 4].double().pow(p32).float()
                kb = utils.bbox_vote(kb.contiguous(), sb.contiguous(), self.bbox_vote_thresh)
            img_boxes.append(kb)
        torch.cuda.synchronize()
        t2 = time.time()
        self.last_timing = {"forward": t1 - t0, "nms": t2 - t1, "total": t2 - t0}  # Tester_FRCNN.lua:130-136
        return img_boxes, (output, bbox_pred)

    def keepTopKPerImage(self, img_boxes, k=100):
        return utils.keep_top_k(img_boxes, k)[0]
