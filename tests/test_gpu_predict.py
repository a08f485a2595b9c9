"""Device inference tail (csrc/predict.hip through vn_rpn_predict / voxelnet_amd.predict / RPN3D.predict) against the
oracle (oracle/predict.py, pinned to the reference by tests/golden/predict_car.npz) and the fixture itself.
Bar: the kept detections — same count, same scores (bit-exact), in the same order; boxes within 2 fp32 ulp of the
reference's (the device's float32 exp / float64 sin, cos may differ from NumPy's in the last bit)."""
import os

import numpy as np
import pytest
import torch

from oracle import predict as op
from oracle import targets as ot

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "predict_car.npz")


def _maps():
    from test_oracle_predict import maps
    return maps()


def test_predict_matches_reference_fixture():
    from voxelnet_amd.predict import BoxDecoder
    g = np.load(GOLD)
    probs, deltas = _maps()
    boxes, scores = BoxDecoder("Car", DEV)(torch.from_numpy(probs).to(DEV), torch.from_numpy(deltas).to(DEV))
    for b in range(int(g["n_samples"])):
        assert np.array_equal(scores[b], g[f"ret_scores{b}"]), b
        assert boxes[b].shape == g[f"ret_boxes{b}"].shape
        np.testing.assert_allclose(boxes[b], g[f"ret_boxes{b}"], rtol=2.4e-7, atol=1e-6)


@pytest.mark.parametrize("seed,dense", [(5, False), (6, True)])
def test_predict_matches_oracle_random(seed, dense):
    """dense: thousands of candidates above the threshold, exact score ties, degenerate boxes"""
    from voxelnet_amd.predict import BoxDecoder
    rng = np.random.default_rng(seed)
    B, h, w = 2, 200, 176
    probs = (rng.random((B, 2, h, w)) * (1.0 if dense else 0.97)).astype(np.float32)
    deltas = (rng.standard_normal((B, 14, h, w)) * 0.3).astype(np.float32)
    if dense:
        probs[0, 0, 3, 5:9] = 1.0                 # ties: the larger flat index first (oracle/predict.py)
        probs[1, 1, 7, 7] = 1.0
    anchors = ot.generate_anchors("Car")
    rb, rs = op.predict_boxes(probs, deltas, anchors)
    boxes, scores = BoxDecoder("Car", DEV)(torch.from_numpy(probs).to(DEV), torch.from_numpy(deltas).to(DEV))
    for b in range(B):
        assert np.array_equal(scores[b], rs[b]), b
        np.testing.assert_allclose(boxes[b], rb[b].reshape(-1, 7), rtol=2.4e-7, atol=1e-6)


def test_rpn3d_predict_returns_the_reference_format():
    from voxelnet_amd import model as M
    probs, deltas = _maps()
    model = M.RPN3D("Car").to(DEV)
    data = (["000001", "000002", "000003"], None, None, None, None, None, None)
    tag, ret = model.predict(data, torch.from_numpy(probs).to(DEV), torch.from_numpy(deltas).to(DEV))
    g = np.load(GOLD)
    assert tag == data[0] and len(ret) == 3
    for b in range(3):
        n = g[f"ret_boxes{b}"].shape[0]
        assert ret[b].shape == (n, 9) if n else ret[b].shape[0] == 0
        if n:
            assert ret[b].dtype.kind == "U" and (ret[b][:, 0] == "Car").all()
            np.testing.assert_allclose(ret[b][:, 8].astype(np.float32), g[f"ret_scores{b}"], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        model.predict(data, torch.from_numpy(probs).to(DEV), torch.from_numpy(deltas).to(DEV), summary=True)
    with pytest.raises(M._lib.VoxelnetHipError):
        model.predict(data, torch.from_numpy(probs), torch.from_numpy(deltas))
