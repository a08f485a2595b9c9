"""oracle/predict.py (NumPy restatement of utils.deltas_to_boxes_3d, model.filter_boxes and utils.nms) against
tests/golden/predict_car.npz, written by tools/gen_golden.py `predict` from the imported reference (CPU tensors).
The decoded boxes must be bit-identical (same NumPy float32/float64 expressions); the kept boxes and scores of every
sample must be identical, in the reference's order."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import predict as op
from oracle import targets as ot

GOLD = os.path.join(os.path.dirname(__file__), "golden", "predict_car.npz")


def maps():
    """the generator's seeded maps, restated (tools/gen_golden.py::predict_maps)"""
    rng = np.random.default_rng(91)
    B, h, w = 3, 200, 176
    probs = (rng.random((B, 2, h, w)) * 0.9).astype(np.float32)
    deltas = (rng.standard_normal((B, 14, h, w)) * 0.15).astype(np.float32)
    for b in range(2):
        for _ in range(12):
            c, y, x = rng.integers(0, 2), rng.integers(2, h - 2), rng.integers(2, w - 8)
            probs[b, c, y, x:x + 6] = (0.96 + 0.04 * rng.random(6)).astype(np.float32)
    probs[0, 1, 10, 10] = np.float32(0.96)
    return probs, deltas


def test_decode_is_bit_identical():
    g = np.load(GOLD)
    probs, deltas = maps()
    boxes = op.deltas_to_boxes_3d(deltas, ot.generate_anchors("Car"))
    assert boxes.dtype == np.float32 and boxes.shape == (3, 70400, 7)
    assert hashlib.sha256(np.ascontiguousarray(boxes).tobytes()).hexdigest() == str(g["boxes_sha"])
    assert np.array_equal(boxes[:, ::997], g["boxes_sample"])


def test_filter_and_nms_match_reference():
    g = np.load(GOLD)
    probs, deltas = maps()
    boxes, scores = op.predict_boxes(probs, deltas, ot.generate_anchors("Car"))
    for b in range(int(g["n_samples"])):
        assert int((probs[b].reshape(-1) >= np.float32(op.SCORE_THRES)).sum()) == int(g[f"n_above{b}"])
        assert np.array_equal(scores[b], g[f"ret_scores{b}"]), b
        assert np.array_equal(boxes[b].reshape(-1, 7), g[f"ret_boxes{b}"]), b
    assert boxes[2].shape[0] == 0


def test_nms_semantics():
    """only the top_k highest scores enter; equal scores: the larger index first; IoU without +1, <= keeps"""
    b = np.array([[0, 0, 2, 2], [0, 0, 2, 2], [10, 10, 12, 12], [0.5, 0.5, 2.5, 2.5]], dtype=np.float64)
    s = np.array([0.99, 0.99, 0.97, 0.98], dtype=np.float32)
    assert list(op.nms(b, s, 0.1, 20)) == [1, 2]          # 1 beats its twin 0 (tie: larger index), 3 overlaps 1
    assert list(op.nms(b, s, 0.1, 1)) == [1]
    assert list(op.nms(np.zeros((0, 4)), np.zeros(0, np.float32), 0.1, 20)) == []
