"""CPU: the oracle's field-of-view crop (oracle/fov.py, preprocess_data.py:42-103) against tests/golden/fov_crop.npz —
the imported reference's align_img_and_velo on every 6th point of a bundled KITTI frame plus points on the decision
boundaries (tools/gen_golden.py fov).  Same NumPy float32 calls: the surviving rows are bit-equal, in order."""
import numpy as np

from oracle import fov as of


def test_oracle_fov_crop(golden):
    g = golden("fov_crop")
    rows, cols = (int(v) for v in g["image_shape"])
    out, idx = of.fov_crop(g["points"], g["P"], g["Tr"], g["R"], rows, cols)
    assert out.dtype == np.float32 and np.array_equal(out, g["kept"])
    assert np.array_equal(g["points"][idx], g["kept"]) and (np.diff(idx) > 0).all()      # input order kept
    assert 0 < out.shape[0] < g["points"].shape[0]
    # the boundary points are really in there: dropped by each of the three tests
    pts = g["points"]
    assert (pts[:, 3] <= 0).sum() > 0 and (pts[:, 0] < 0).sum() > 0
