"""the oracle's wire-format restatements on hand-worked cases (DataSetJSON.lua:157-233, testCoco/init.lua:69-85, utils.lua:335-372)"""
import numpy as np


def test_filter_area_score_permute_hand_case(O):
    # rows {y1,x1,y2,x2}; areas (y2-y1)*(x2-x1): 6, 2, 12, 0.25, 20
    b = np.array([[1, 1, 3, 4], [2, 2, 3, 4], [5, 1, 8, 5], [1, 1, 1.5, 1.5], [10, 10, 14, 15]], np.float32)
    s = np.array([0.2, 0.9, 0.5, 0.99, 0.5], np.float32)
    fb, fs = O.filter_area(b, s, 2.0)                      # strictly greater (DataSetJSON.lua:178 `gt`): the area-2 row goes too
    assert np.array_equal(fb, b[[0, 2, 4]]) and np.array_equal(fs, s[[0, 2, 4]])
    assert O.filter_area(b, s, 0)[0] is not None and O.filter_area(b, s, 0)[0].shape == (5, 4)  # area == 0: untouched (line 172)
    pb, ps = O.prepare_proposals(b, s, min_area=2.0, best_number=2)
    # survivors rows 0, 2, 4 with scores .2, .5, .5 -> best 2 = rows 2, 4 (tie: lower row first), columns {2,1,4,3}
    assert np.array_equal(pb, np.array([[1, 5, 5, 8], [10, 10, 15, 14]], np.float32)) and np.array_equal(ps, np.array([0.5, 0.5], np.float32))
    pb, ps = O.prepare_proposals(b, s, min_area=2.0, best_number=3)   # not more than best_number rows: no sort (line 161)
    assert np.array_equal(pb, b[[0, 2, 4]][:, [1, 0, 3, 2]]) and np.array_equal(ps, s[[0, 2, 4]])
    assert O.prepare_proposals(b)[1] is None


def test_coco_rows_and_results_table_hand_case(O):
    d = np.array([[11, 21, 31, 61, 0.75, 2], [1, 1, 5, 9, 0.5, 1]], np.float32)
    rows = O.coco_rows(d, 139, [18.0, 44.0])
    assert np.array_equal(rows, np.array([[139, 10, 20, 20, 40, 0.75, 44], [139, 0, 0, 4, 8, 0.5, 18]], np.float32))
    ab = [[d[:1, :5], np.zeros((0, 5), np.float32)], [None, d[1:, :5]]]
    t = O.save_results_table(ab, "toy")
    assert t["images"].tolist() == [1, 2] and t["detections"]["categories"].tolist() == [1, 2] and t["detections"]["images"].tolist() == [1, 2]
    assert np.array_equal(t["detections"]["boxes"], d[:, :4]) and np.array_equal(t["detections"]["scores"], d[:, 4])
