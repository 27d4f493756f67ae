"""Torch7 binary serialisation (multipathnet_amd/t7.py) — the `.t7` wire format of proposal tables in and boxes / results out
(DataSetJSON.lua:131,150; run_test.lua:62,75,82; utils.lua:335-372).  CPU only: a hand-built byte fixture written straight from
torch7's File.lua / Tensor.c layout, round trips, shared references, strided views, tds containers, and the proposal-merge rules."""
import os
import struct
import tempfile

import numpy as np
import pytest

from multipathnet_amd import t7


def _i(v):
    return struct.pack("<i", v)


def _q(v):
    return struct.pack("<q", v)


def _s(s):
    return _i(len(s)) + s.encode()


def _fixture_bytes():
    """torch.save('p.t7', {boxes = {torch.FloatTensor{{1,2,3,4},{5,6,7,8}}}, images = {'a.jpg'}, n = 2, ok = true}) spelled out
    byte group by byte group (table keys in this insertion order)."""
    b = b""
    b += _i(3) + _i(1) + _i(4)                               # TABLE, index 1, 4 pairs
    b += _i(2) + _s("boxes")                                 # key: STRING
    b += _i(3) + _i(2) + _i(1)                               #   TABLE, index 2, 1 pair
    b += _i(1) + struct.pack("<d", 1.0)                      #     key 1 (NUMBER)
    b += _i(4) + _i(3) + _s("V 1") + _s("torch.FloatTensor")  #     TORCH, index 3
    b += _i(2) + _q(2) + _q(4) + _q(4) + _q(1) + _q(1)       #       nDim 2, size {2,4}, stride {4,1}, storageOffset 1
    b += _i(4) + _i(4) + _s("V 1") + _s("torch.FloatStorage")  #     its storage: TORCH, index 4
    b += _q(8) + np.arange(1, 9, dtype="<f4").tobytes()
    b += _i(2) + _s("images")
    b += _i(3) + _i(5) + _i(1) + _i(1) + struct.pack("<d", 1.0) + _i(2) + _s("a.jpg")
    b += _i(2) + _s("n") + _i(1) + struct.pack("<d", 2.0)
    b += _i(2) + _s("ok") + _i(5) + _i(1)
    return b


def test_read_hand_built_fixture():
    v = t7.loads(_fixture_bytes())
    assert set(v) == {"boxes", "images", "n", "ok"}
    assert isinstance(v["boxes"], list) and v["boxes"][0].dtype == np.float32
    assert np.array_equal(v["boxes"][0], np.arange(1, 9, dtype=np.float32).reshape(2, 4))
    assert v["images"] == ["a.jpg"] and v["n"] == 2.0 and v["ok"] is True


def test_writer_emits_the_fixture_bytes():
    obj = {"boxes": [np.arange(1, 9, dtype=np.float32).reshape(2, 4)], "images": ["a.jpg"], "n": 2, "ok": True}
    assert t7.dumps(obj) == _fixture_bytes()


def test_round_trip_types_and_shapes():
    rng = np.random.default_rng(0)
    obj = {"f": rng.random((3, 5)).astype(np.float32), "d": rng.random(7), "l": np.arange(6, dtype=np.int64).reshape(2, 3),
           "b": np.array([1, 2, 255], np.uint8), "i": np.array([[1], [2]], np.int32), "empty": np.zeros((0,), np.float32),
           "nested": {"x": [1.5, "s", None, False], 3: "three"}, "hash": t7.TdsHash({1: np.ones((2, 5), np.float32), 2: np.zeros((0,), np.float32)}),
           "vec": t7.TdsVec(["a", 2.0])}
    back = t7.loads(t7.dumps(obj))
    for k in ("f", "d", "l", "b", "i"):
        assert back[k].dtype == obj[k].dtype and np.array_equal(back[k], obj[k])
    assert back["empty"].size == 0
    assert back["nested"]["x"][:2] == [1.5, "s"] and back["nested"]["x"][3] is False and back["nested"][3.0] == "three"
    assert isinstance(back["hash"], t7.TdsHash) and np.array_equal(back["hash"][1.0], np.ones((2, 5), np.float32)) and back["hash"][2.0].size == 0
    assert isinstance(back["vec"], t7.TdsVec) and list(back["vec"]) == ["a", 2.0]


def test_strided_views_and_shared_references():
    """two tensors over ONE storage (a transposed view and a narrow() with an offset), and a table referenced twice"""
    st = np.arange(12, dtype="<f4")
    storage = _i(4) + _i(9) + _s("V 1") + _s("torch.FloatStorage") + _q(12) + st.tobytes()
    tens_t = _i(4) + _i(7) + _s("V 1") + _s("torch.FloatTensor") + _i(2) + _q(4) + _q(3) + _q(1) + _q(4) + _q(1) + storage  # 4x3, strides {1,4}
    tens_n = _i(4) + _i(8) + _s("V 1") + _s("torch.FloatTensor") + _i(1) + _q(3) + _q(1) + _q(6) + _i(4) + _i(9)          # narrow: offset 6 (1-based), storage by reference
    shared = _i(3) + _i(5) + _i(1) + _i(2) + _s("k") + _i(1) + struct.pack("<d", 9.0)
    b = _i(3) + _i(1) + _i(4)
    b += _i(1) + struct.pack("<d", 1.0) + tens_t
    b += _i(1) + struct.pack("<d", 2.0) + tens_n
    b += _i(1) + struct.pack("<d", 3.0) + shared
    b += _i(1) + struct.pack("<d", 4.0) + _i(3) + _i(5)      # the same table again: index only
    v = t7.loads(b)
    assert np.array_equal(v[0], st.reshape(3, 4).T)
    assert np.array_equal(v[1], st[5:8])
    assert v[2] == {"k": 9.0} and v[3] is v[2]
    a = np.ones((2, 2), np.float32)
    back = t7.loads(t7.dumps([a, a, {"t": a}]))   # the writer shares too: one tensor object, three references
    assert back[0] is back[1] and back[2]["t"] is back[0]


def test_errors():
    with pytest.raises(t7.T7Error):
        t7.loads(_fixture_bytes()[:40])
    with pytest.raises(t7.T7Error):
        t7.loads(_i(6) + _i(1))      # a Lua function
    with pytest.raises(t7.T7Error):
        t7.dumps({"x": object()})


def test_boxes_and_results_files_and_proposal_merge():
    from multipathnet_amd import formats
    import torch
    rng = np.random.default_rng(1)
    aboxes = [[torch.from_numpy(rng.random((3, 5)).astype(np.float32)), torch.zeros((0, 5))], [None, torch.from_numpy(rng.random((2, 5)).astype(np.float32))]]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "boxes.t7")
        formats.save_boxes(p, aboxes)
        back = formats.load_boxes(p)
        assert len(back) == 2 and back[0][1].shape == (0, 5) and back[1][0].shape == (0, 5)
        assert np.array_equal(back[0][0], aboxes[0][0].numpy()) and np.array_equal(back[1][1], aboxes[1][1].numpy())
        r = os.path.join(d, "results.t7")
        formats.save_results_file(r, aboxes, "toy")
        res = t7.load(r)
        assert res["dataset"] == "toy" and res["images"].tolist() == [1, 2]
        assert res["detections"]["boxes"].shape == (5, 4) and res["detections"]["categories"].tolist() == [1, 1, 1, 2, 2]
        assert res["detections"]["images"].tolist() == [1, 1, 1, 2, 2]
        # loadAndMergeProposals (DataSetJSON.lua:124-160): per image name, boxes / scores concatenated in file order; a file without scores scores 0
        f1, f2 = os.path.join(d, "p1.t7"), os.path.join(d, "p2.t7")
        b1 = [rng.random((4, 4)).astype(np.float32), rng.random((2, 4)).astype(np.float32)]
        b2 = [rng.random((3, 4)).astype(np.float32), rng.random((1, 4)).astype(np.float32)]
        t7.save(f1, {"boxes": b1, "scores": [rng.random(4).astype(np.float32), rng.random(2).astype(np.float32)], "images": ["a.jpg", "b.jpg"]})
        t7.save(f2, {"boxes": b2, "images": ["b.jpg", "c.jpg"]})
        one = formats.load_proposals(f1)
        assert one["images"] == ["a.jpg", "b.jpg"] and np.array_equal(one["boxes"][1], b1[1])
        m = formats.load_proposals([f1, f2])
        assert m["images"] == ["a.jpg", "b.jpg", "c.jpg"]
        assert np.array_equal(m["boxes"][1], np.concatenate([b1[1], b2[0]])) and m["scores"][1][2:].tolist() == [0, 0, 0]
        assert np.array_equal(m["boxes"][2], b2[1]) and m["scores"][2].tolist() == [0]


def _tensor_bytes(size, stride, offset, n_storage, claim=None):
    """a lone torch.FloatTensor record with the given (unchecked-by-the-format) geometry over an n_storage-element storage;
    `claim` = the element count the storage header states (default n_storage)."""
    b = _i(4) + _i(1) + _s("V 1") + _s("torch.FloatTensor")
    b += _i(len(size)) + b"".join(_q(v) for v in size) + b"".join(_q(v) for v in stride) + _q(offset)
    b += _i(4) + _i(2) + _s("V 1") + _s("torch.FloatStorage")
    b += _q(n_storage if claim is None else claim) + np.arange(n_storage, dtype="<f4").tobytes()
    return b


def test_malformed_tensor_geometry_raises_t7error():
    """sizes / strides / offset / storage counts come unchecked from the file; a view that leaves its storage must
    raise T7Error instead of reading out of bounds (a 1000x1000 tensor over a 4-element storage used to load as garbage)."""
    ok = t7.loads(_tensor_bytes([2, 2], [2, 1], 1, 4))
    assert np.array_equal(ok, np.arange(4, dtype=np.float32).reshape(2, 2))
    assert np.array_equal(t7.loads(_tensor_bytes([2, 3], [0, 1], 2, 4)), np.array([[1, 2, 3], [1, 2, 3]], np.float32))  # expand()ed view
    for size, stride, off in [([1000, 1000], [1000, 1], 1), ([2, 2], [2, 1], 2), ([2, 2], [4, 1], 1), ([4], [1], 0), ([2], [-1], 2),
                              ([-1], [1], 1), ([5], [1], 1)]:
        with pytest.raises(t7.T7Error):
            t7.loads(_tensor_bytes(size, stride, off, 4))
    for claim in (-1, 5, 2 ** 40):
        with pytest.raises(t7.T7Error):
            t7.loads(_tensor_bytes([2], [1], 1, 4, claim=claim))
    assert t7.loads(_tensor_bytes([0, 4], [4, 1], 1, 4)).shape == (0, 4)


def test_proposal_merge_with_an_empty_per_image_tensor():
    """TableConcat (DataSetJSON.lua:114-122) returns the other operand when one side is empty — an image without
    proposals in one of the merged files is a 0-element tensor, shape (0,), which does not concatenate with (n,4)."""
    from multipathnet_amd import formats
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as d:
        f1, f2, f3 = (os.path.join(d, n) for n in ("p1.t7", "p2.t7", "p3.t7"))
        b1 = [np.zeros((0,), np.float32), rng.random((2, 4)).astype(np.float32)]
        s1 = [np.zeros((0,), np.float32), rng.random(2).astype(np.float32)]
        b2 = [rng.random((3, 4)).astype(np.float32), np.zeros((0,), np.float32)]
        s2 = [rng.random(3).astype(np.float32), np.zeros((0,), np.float32)]
        t7.save(f1, {"boxes": b1, "scores": s1, "images": ["a.jpg", "b.jpg"]})
        t7.save(f2, {"boxes": b2, "scores": s2, "images": ["a.jpg", "b.jpg"]})
        t7.save(f3, {"boxes": [np.zeros((0,), np.float32)], "images": ["a.jpg"]})   # unscored AND empty
        m = formats.load_proposals([f1, f2, f3])
        assert np.array_equal(m["boxes"][0], b2[0]) and np.array_equal(m["scores"][0], s2[0])   # empty first, then 3 boxes, then empty
        assert np.array_equal(m["boxes"][1], b1[1]) and np.array_equal(m["scores"][1], s1[1])
        m2 = formats.load_proposals([f2, f1])
        assert np.array_equal(m2["boxes"][0], b2[0]) and np.array_equal(m2["boxes"][1], b1[1])
