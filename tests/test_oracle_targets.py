"""oracle/targets.py (NumPy restatement of utils.generate_anchors / label_to_gt_box_3d / bbox_iou / generate_targets)
against tests/golden/targets_car.npz, written by tools/gen_golden.py from the imported reference.
Integer work (which anchors are positive / negative) is bit-exact; regression targets are float64 on both sides and
must agree to 1e-12."""
import hashlib
import os

import numpy as np

from oracle import targets as ot

GOLD = os.path.join(os.path.dirname(__file__), "golden", "targets_car.npz")


def load():
    g = np.load(GOLD, allow_pickle=False)
    n = int(g["n_samples"])
    labels = [[str(s) for s in g[f"labels{b}"]] for b in range(n)]
    return g, n, labels


def test_anchors_are_bit_identical():
    g, _, _ = load()
    a = ot.generate_anchors("Car")
    assert a.shape == (200, 176, 2, 7) and a.dtype == np.float64
    assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g["anchors_sha"])
    assert np.array_equal(a[::50, ::44], g["anchors_corner"])


def test_label_parsing_and_camera_to_lidar():
    g, n, labels = load()
    gt = ot.label_to_gt_box_3d(labels, "Car", "lidar")
    for b in range(n):
        assert gt[b].shape == g[f"gt{b}"].shape
        np.testing.assert_allclose(gt[b], g[f"gt{b}"], rtol=0, atol=1e-12)


def test_targets_match_reference():
    g, n, labels = load()
    shape = tuple(int(v) for v in g["shape"])
    pos, neg, tgt = ot.generate_targets(labels, shape, ot.generate_anchors("Car"))
    assert pos.shape == (n, *shape, 2) and neg.shape == (n, *shape, 2) and tgt.shape == (n, *shape, 14)
    for b in range(n):
        assert np.array_equal(np.flatnonzero(pos[b]).astype(np.int32), g[f"pos_idx{b}"]), b
        assert np.array_equal(np.packbits(neg[b].reshape(-1).astype(np.uint8)), g[f"neg_bits{b}"]), b
        nz = np.flatnonzero(tgt[b])
        assert np.array_equal(nz.astype(np.int32), g[f"tgt_idx{b}"]), b
        np.testing.assert_allclose(tgt[b].reshape(-1)[nz], g[f"tgt_val{b}"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose([pos[b].sum(), neg[b].sum(), np.abs(tgt[b]).sum()], g[f"sums{b}"], rtol=1e-12)
    # the sample without accepted objects: every anchor negative, none positive (utils.py:416 on an (N,0) IoU matrix)
    assert pos[2].sum() == 0 and neg[2].sum() == shape[0] * shape[1] * 2


def test_iou_quirks():
    """zero-extent anchor boxes (utils.py:219-225) and the y1-x1 union term (utils.py:367)"""
    a = ot.anchor_standup_2d(ot.generate_anchors("Car"))
    assert np.array_equal(a[:, 0], a[:, 2]) and np.array_equal(a[:, 1], a[:, 3])
    box1 = np.array([[10.0, 2.0, 10.0, 2.0]], dtype=np.float32)
    box2 = np.array([[8.0, 1.0, 12.0, 3.0], [50.0, 1.0, 52.0, 3.0]], dtype=np.float32)
    iou = ot.bbox_iou(box1, box2)
    ua = (2.0 - 10.0 + 1.0) * (2.0 - 2.0 + 1.0) + 5.0 * 3.0 - 1.0
    assert iou.dtype == np.float32 and iou[0, 0] == np.float32(1.0 / ua) and iou[0, 1] == 0


def test_product_host_helpers_match_oracle():
    """voxelnet_amd.targets' host half (label parsing, camera->lidar, stand-up rectangles, anchors) is NumPy and runs
    without a GPU: bit-identical to the oracle's restatement on the fixture's labels."""
    from voxelnet_amd import targets as T
    g, n, labels = load()
    for a, b in zip(T.label_to_gt_box_3d(labels, "Car"), ot.label_to_gt_box_3d(labels, "Car")):
        assert np.array_equal(a, b)
        assert np.array_equal(T.gt_standup_boxes(a), ot.gt_standup_2d(b))
    for cls in ("Car", "Pedestrian", "Cyclist"):
        assert np.array_equal(T.generate_anchors(cls), ot.generate_anchors(cls))
