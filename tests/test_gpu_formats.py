"""wire formats either side of the hot path: proposal tables (DataSetJSON.lua:157-239) and COCO result rows
(testCoco/init.lua:65-85, utils.lua:335-372), checked against numpy restatements of the cited Lua lines."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prepare_proposals(dev):
    from multipathnet_amd import formats
    rng = np.random.default_rng(0)
    yx = rng.uniform(1, 500, (300, 2)).astype(np.float32)
    b = np.concatenate([yx, yx + rng.uniform(0, 40, (300, 2)).astype(np.float32)], 1)  # {y1,x1,y2,x2}
    s = rng.random(300).astype(np.float32)
    out, sc = formats.prepare_proposals(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), min_area=2.0, best_number=100)
    # Lua: wh = b[:,3:4] - b[:,1:2]; keep wh1*wh2 > area; then best_number by score; then index(2, {2,1,4,3})
    wh = b[:, 2:] - b[:, :2]
    keep = np.nonzero(wh[:, 0] * wh[:, 1] > np.float32(2.0))[0]
    bb, ss = b[keep], s[keep]
    order = np.argsort(-ss, kind="stable")[:100]
    assert out.shape == (100, 4)
    assert np.array_equal(np.sort(sc.cpu().numpy())[::-1], np.sort(ss[order])[::-1])
    exp = bb[order][:, [1, 0, 3, 2]]
    got = out.cpu().numpy()
    assert np.array_equal(got[np.lexsort(got.T)], exp[np.lexsort(exp.T)])  # order among equal scores is unspecified in torch.sort
    out2, sc2 = formats.prepare_proposals(torch.from_numpy(b).to(dev), None, min_area=0.0)
    assert np.array_equal(out2.cpu().numpy(), b[:, [1, 0, 3, 2]]) and sc2 is None
    assert formats.prepare_proposals(torch.zeros((0, 4), device=dev))[0].shape == (0, 4)


def test_detections_to_coco_rows(dev):
    from multipathnet_amd import formats
    rng = np.random.default_rng(1)
    d = np.concatenate([rng.uniform(1, 300, (50, 2)), rng.uniform(301, 600, (50, 2)), rng.random((50, 1)), rng.integers(1, 21, (50, 1))], 1).astype(np.float32)
    cats = [float(100 + 3 * c) for c in range(20)]
    rows = formats.detections_to_coco(torch.from_numpy(d).to(dev), torch.tensor([37], dtype=torch.int32, device=dev), 4242.0, cats)
    exp = np.stack([np.full(37, 4242.0, np.float32), d[:37, 0] - 1, d[:37, 1] - 1, d[:37, 2] - d[:37, 0], d[:37, 3] - d[:37, 1], d[:37, 4],
                    np.array([cats[int(c) - 1] for c in d[:37, 5]], np.float32)], 1)
    assert np.array_equal(rows.cpu().numpy(), exp)
    res = formats.save_results([[torch.from_numpy(d[:3, :5]), torch.zeros((0, 5))], [None, torch.from_numpy(d[3:5, :5])]], "toy")
    assert res["detections"]["boxes"].shape == (5, 4) and res["detections"]["categories"].tolist() == [1, 1, 1, 2, 2]
    assert res["detections"]["images"].tolist() == [1, 1, 1, 2, 2] and res["images"].tolist() == [1, 2]
