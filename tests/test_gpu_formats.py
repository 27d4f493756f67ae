"""wire formats either side of the hot path: proposal tables (DataSetJSON.lua:157-239) and COCO result rows
(testCoco/init.lua:65-85, utils.lua:335-372), checked against the oracle module's restatements of the cited Lua lines (oracle/mpn_oracle.py: filter_area, filter_score,
prepare_proposals, coco_rows, save_results_table)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prepare_proposals(O, dev):
    """device path == the oracle module's restatement of DataSetJSON.lua:157-233 (filterArea -> filterScore -> permute), row for
    row and in order — with distinct scores, and with tied scores under the documented stable tie order"""
    from multipathnet_amd import formats
    rng = np.random.default_rng(0)
    yx = rng.uniform(1, 500, (300, 2)).astype(np.float32)
    b = np.concatenate([yx, yx + rng.uniform(0, 40, (300, 2)).astype(np.float32)], 1)  # {y1,x1,y2,x2}
    b[7, 2:] = b[7, :2] + 1.0                                                           # area exactly 1 < 2: dropped
    for tied in (False, True):
        s = rng.random(300).astype(np.float32)
        if tied:
            s = np.round(s * 8) / np.float32(8)                                         # 9 distinct values: runs of ~33 equal scores
        out, sc = formats.prepare_proposals(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), min_area=2.0, best_number=100)
        eb, es = O.prepare_proposals(b, s, min_area=2.0, best_number=100)
        assert eb.shape == (100, 4)
        assert np.array_equal(out.cpu().numpy(), eb) and np.array_equal(sc.cpu().numpy(), es)
    # fewer rows than best_number: no sort at all, row order kept (DataSetJSON.lua:161)
    out, sc = formats.prepare_proposals(torch.from_numpy(b[:50]).to(dev), torch.from_numpy(s[:50]).to(dev), min_area=2.0, best_number=100)
    eb, es = O.prepare_proposals(b[:50], s[:50], min_area=2.0, best_number=100)
    assert np.array_equal(out.cpu().numpy(), eb) and np.array_equal(sc.cpu().numpy(), es)
    out2, sc2 = formats.prepare_proposals(torch.from_numpy(b).to(dev), None, min_area=0.0)
    assert np.array_equal(out2.cpu().numpy(), O.prepare_proposals(b)[0]) and sc2 is None
    assert formats.prepare_proposals(torch.zeros((0, 4), device=dev))[0].shape == (0, 4)


def test_detections_to_coco_rows(O, dev):
    from multipathnet_amd import formats
    rng = np.random.default_rng(1)
    d = np.concatenate([rng.uniform(1, 300, (50, 2)), rng.uniform(301, 600, (50, 2)), rng.random((50, 1)), rng.integers(1, 21, (50, 1))], 1).astype(np.float32)
    cats = [float(100 + 3 * c) for c in range(20)]
    rows = formats.detections_to_coco(torch.from_numpy(d).to(dev), torch.tensor([37], dtype=torch.int32, device=dev), 4242.0, cats)
    assert np.array_equal(rows.cpu().numpy(), O.coco_rows(d[:37], 4242.0, cats))
    ab = [[torch.from_numpy(d[:3, :5]), torch.zeros((0, 5))], [None, torch.from_numpy(d[3:5, :5])]]
    res = formats.save_results(ab, "toy")
    exp = O.save_results_table([[d[:3, :5], np.zeros((0, 5), np.float32)], [None, d[3:5, :5]]], "toy")
    assert res["dataset"] == exp["dataset"] and np.array_equal(res["images"].numpy(), exp["images"])
    for k in ("boxes", "scores", "categories", "images"):
        assert np.array_equal(res["detections"][k].numpy(), exp["detections"][k]), k
