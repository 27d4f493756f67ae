"""Pins the oracle's NMS family: (1) the reference's IoU known-answer vector (test.lua:40-52);
(2) bit-for-bit agreement of the C restatement with the reference's OWN nms.c compiled unmodified
(oracle/_ref/libnms_ref.so) over score regimes that exercise the tie-break history (nms.c:74-98)."""
import numpy as np
import pytest

from conftest import case_seed, random_scored_boxes


def test_iou_known_answer(O):
    # test.lua:40-52 — utils.boxoverlap KAT, tolerance 5e-3 as in the reference
    a = np.array([[0, 0, 100, 100], [0, 50, 100, 150], [50, 0, 150, 100], [50, 50, 150, 150], [100, 100, 200, 200]], np.float32)
    gt = np.array([1 / 7, 1 / 3, 1 / 3, 1, 1 / 7], np.float32)
    assert np.abs(O.boxoverlap(a, [50, 50, 150, 150]) - gt).max() < 5e-3
    if O.have_ref():
        got = np.array([O.ref_overlap(r, [50, 50, 150, 150]) for r in a], np.float32)
        assert np.array_equal(got, O.boxoverlap(a, [50, 50, 150, 150]))


def test_overlap_degenerate(O):
    assert O.overlap([0, 0, 10, 10], [20, 20, 30, 30]) == 0.0           # disjoint
    assert O.overlap([0, 0, 10, 10], [11, 0, 20, 10]) == 0.0            # w == 0 exactly (x2-x1+1 = 0)
    assert O.overlap([0, 0, 10, 10], [10, 10, 20, 20]) > 0.0            # touching pixels overlap by 1 px (+1 convention)


@pytest.mark.parametrize("regime", ["distinct", "ties", "saturated", "allequal"])
@pytest.mark.parametrize("n", [1, 2, 7, 64, 65, 300, 1000])
@pytest.mark.parametrize("thr", [0.3, 0.5])
def test_nms_matches_reference_c(O, regime, n, thr):
    if not O.have_ref():
        pytest.skip("oracle/_ref/libnms_ref.so not built (needs /root/reference)")
    rng = np.random.default_rng(case_seed(regime, n))
    sb = random_scored_boxes(rng, n, regime, span=300.0 if n <= 65 else 1000.0)
    mine, idx = O.nms(sb, thr, return_index=True)
    ref = O.ref_nms(sb, thr)
    assert mine.shape == ref.shape and np.array_equal(mine, ref)
    assert np.array_equal(sb[idx], mine)


def test_nms_empty_and_single(O):
    assert O.nms(np.zeros((0, 5), np.float32), 0.3).shape == (0, 5)
    if O.have_ref():
        assert O.ref_nms(np.zeros((0, 5), np.float32), 0.3).shape == (0, 5)
    one = np.array([[1, 1, 5, 5, 0.3]], np.float32)
    assert np.array_equal(O.nms(one, 0.3), one)


def test_nms_tie_history_is_order_dependent(O):
    """The property that forces the position-key emulation: with equal scores the winner is NOT simply
    'lowest index' once a swap has moved the old first element (nms.c:83-85)."""
    if not O.have_ref():
        pytest.skip("needs reference nms.c")
    sb = np.array([[0, 0, 10, 10, 0.5],       # A (first)
                   [100, 100, 110, 110, 0.5],  # B
                   [200, 200, 210, 210, 0.9],  # C best -> swapped with A, so A now sits after B
                   ], np.float32)
    ref = O.ref_nms(sb, 0.3)
    assert np.array_equal(ref[:, :4], sb[[2, 1, 0], :4])  # C, then B (now first), then A
    assert np.array_equal(O.nms(sb, 0.3), ref)


@pytest.mark.parametrize("regime", ["distinct", "ties"])
def test_bbox_vote_matches_reference_c(O, regime):
    if not O.have_ref():
        pytest.skip("needs reference nms.c")
    rng = np.random.default_rng(11)
    sb = random_scored_boxes(rng, 400, regime, span=400.0)
    sb[:, 4] = np.maximum(sb[:, 4], 1e-3)
    keep = O.nms(sb, 0.3)
    a, b = O.bbox_vote(keep, sb, 0.5), O.ref_bbox_vote(keep, sb, 0.5)
    assert np.array_equal(a, b)


def test_decode_roundtrip_property(O):
    # test.lua:17-38 — convertFrom(convertTo(b,t)) == t (we check it in fp32 with a matching tolerance)
    rng = np.random.default_rng(3)
    A, B = rng.random((50, 2)) * 100, rng.random((50, 2)) * 100
    bbox = np.concatenate([A, A + rng.integers(1, 41, (50, 2))], 1).astype(np.float32)
    tbox = np.concatenate([B, B + rng.integers(1, 41, (50, 2))], 1).astype(np.float32)
    # utils.convertTo (utils.lua:176-199): dx=(xc_t-xc)/w, dw=log(w_t/w)
    w, h = bbox[:, 2] - bbox[:, 0], bbox[:, 3] - bbox[:, 1]
    xc, yc = (bbox[:, 0] + bbox[:, 2]) * 0.5, (bbox[:, 1] + bbox[:, 3]) * 0.5
    wt, ht = tbox[:, 2] - tbox[:, 0], tbox[:, 3] - tbox[:, 1]
    xt, yt = (tbox[:, 0] + tbox[:, 2]) * 0.5, (tbox[:, 1] + tbox[:, 3]) * 0.5
    d = np.stack([(xt - xc) / w, (yt - yc) / h, np.log(wt / w), np.log(ht / h)], 1).astype(np.float32)
    out = O.bbox_decode(bbox, d)
    assert np.abs(out - tbox).max() < 1e-3


def test_nms_dense_restatement(O):
    """utils.nms_dense (utils.lua:402-462): hand-worked cases of the index-returning NMS demo.lua uses."""
    assert O.nms_dense(np.zeros((0, 5), np.float32), 0.3).shape == (0,)                       # utils.lua:405-407
    # the higher-scored of two heavily overlapping boxes is picked first and suppresses the other; a far box survives; 1-based picks
    hb = np.array([[10, 10, 50, 50, 0.5], [12, 12, 52, 52, 0.9], [100, 100, 120, 120, 0.1]], np.float32)
    assert O.nms_dense(hb, 0.3).tolist() == [2, 3]
    assert O.nms_dense(hb, 0.95).tolist() == [2, 1, 3]                                       # threshold above their IoU: both stay, score order
    # strict '>' (utils.lua:449 ol:gt(overlap)): IoU == overlap exactly does not suppress.  Two 2x1 boxes sharing one pixel column:
    # inter = 1*... a: x 0..1, b: x 1..2, y 0..0 -> w = 1, h = 1, inter = 1, areas 2 and 2, union 3 -> IoU = 1/3
    eq = np.array([[0, 0, 1, 0, 0.9], [1, 0, 2, 0, 0.8]], np.float32)
    third = np.float32(1) / np.float32(3)
    assert O.nms_dense(eq, float(third)).tolist() == [1, 2] and O.nms_dense(eq, float(np.nextafter(third, np.float32(0)))).tolist() == [1]
    # ties: restated as ascending index (TH's quicksort order among equal scores is unpinned)
    t = np.array([[0, 0, 10, 10, 0.5], [100, 100, 110, 110, 0.5], [200, 200, 210, 210, 0.5]], np.float32)
    assert O.nms_dense(t, 0.3).tolist() == [1, 2, 3]


@pytest.mark.parametrize("n", [1, 7, 64, 300, 1000])
def test_nms_dense_agrees_with_nms_c_on_distinct_scores(O, n):
    """With distinct scores both NMS forms walk the boxes in descending score order and differ only in how the IoU is rounded
    (area-based vs nms.c's); on these inputs no IoU sits within an ulp of the threshold, so the picks index exactly the rows
    nms.c keeps."""
    sb = random_scored_boxes(np.random.default_rng(case_seed("distinct", n, salt=3)), n, "distinct", span=300.0 if n <= 64 else 1000.0)
    picks = O.nms_dense(sb, 0.3)
    ref, ridx = O.nms(sb, 0.3, return_index=True)
    assert np.array_equal(picks - 1, ridx) and np.array_equal(sb[picks - 1], ref)


def test_fused_kernel_model_matches_the_compiled_reference(O):
    """tools/models/nms_fused_model.py is the algorithm csrc/nms.hip's nms_fused_kernel runs for classes with bit-equal scores (alive-by-rank
    and occupied-by-slot bitsets, pos / owner, lazily dropped dead slots, the run's cached slots): both of its forms must reproduce the
    reference's compiled nms.c pick for pick (nms.c:74-98) — on CPU, so that the rule the kernel implements stays pinned without a GPU."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nms_fused_model", os.path.join(root, "tools", "models", "nms_fused_model.py"))
    mdl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mdl)
    from conftest import random_scored_boxes
    n = 0
    for regime in ("distinct", "ties", "saturated", "allequal"):
        for m in (1, 2, 17, 65, 90):
            rng = np.random.default_rng(m * 31 + len(regime))
            sb = random_scored_boxes(rng, m, regime, span=200.0, lo=16.0, hi=150.0)
            ref = O.ref_nms(sb, 0.3) if O.have_ref() else O.nms(sb, 0.3)
            for fn in (mdl.fused, mdl.fused_v2):
                got, idx = fn(sb, 0.3)
                assert np.array_equal(got, ref), (fn.__name__, regime, m)
                assert np.array_equal(sb[idx], ref)
            n += 1
    assert n == 20


def test_lazy_replay_model_matches_the_compiled_reference(O):
    """tools/models/nms_lazy_replay_model.py is the chunked scan with the lazy position replay — nms_scan_kernel's form (V2=False: a move
    landing inside the 64-slot window ends the batch, the slot model is brought up to date before every exact-rule pick) and
    nms_fused_kernel's (V2=True: round-space batches that resolve such landings inside the fixpoint, once per equal-score RUN).  Both must
    reproduce the reference's compiled nms.c pick for pick, incl. on tables with long runs (where the once-per-run invariant is exercised)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nms_lazy_replay_model", os.path.join(root, "tools", "models", "nms_lazy_replay_model.py"))
    mdl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mdl)
    from conftest import random_scored_boxes
    n = 0
    for seed in range(12):
        rng = np.random.default_rng(4200 + seed)
        m = int(rng.choice([5, 64, 65, 97, 130, 200]))
        sb = random_scored_boxes(rng, m, "distinct", span=float(rng.choice([150.0, 300.0])), lo=16.0, hi=150.0)
        if seed % 3 == 0:       # a few tied pairs and a duplicated proposal
            for _ in range(3):
                a, b = rng.choice(m, 2, replace=False)
                sb[b, 4] = sb[a, 4]
            a, b = rng.choice(m, 2, replace=False)
            sb[b] = sb[a]
        else:                   # scores on a few levels: runs of many equal scores
            levels = int(rng.choice([2, 5, 17]))
            sb[:, 4] = (np.round(sb[:, 4] * levels) / levels).astype(np.float32)
        ref = O.ref_nms(sb, 0.3) if O.have_ref() else O.nms(sb, 0.3)
        for v2 in (False, True):
            st = {}
            got, idx = mdl.model_nms(sb, 0.3, st, V2=v2)
            assert np.array_equal(got, ref), (seed, m, v2)
            assert np.array_equal(sb[idx], ref)
        n += 1
    assert n == 12
