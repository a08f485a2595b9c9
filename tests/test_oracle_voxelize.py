"""CPU: the voxelizer oracle (oracle/voxelize_ref.c + numpy twin) against golden
vectors produced by the imported reference (utils.py:10-100).  Bit-exact."""
import hashlib

import numpy as np
import pytest

from oracle import voxelize as ov
from voxelnet_amd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("tag,target", [("car", "Car"), ("ped", "Pedestrian")])
@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_small_clouds_bit_exact(golden, tag, target, impl):
    g = golden(f"voxelize_{tag}_small")
    fn = ov.voxelize if impl == "c" else ov.voxelize_numpy
    with np.errstate(all="ignore"):
        out = fn(g["points"], target)
    assert np.array_equal(out["coordinate_buffer"], g["coordinate_buffer"])
    assert out["coordinate_buffer"].dtype == np.int64
    assert np.array_equal(out["number_buffer"], g["number_buffer"])
    assert out["feature_buffer"].dtype == np.float32
    # bit-exact, including the float64 centroid division (utils.py:87-88)
    assert np.array_equal(out["feature_buffer"].view(np.uint32), g["feature_buffer"].view(np.uint32))
    assert (g["number_buffer"] == (35 if target == "Car" else 45)).sum() >= 2   # overflow exercised


def test_degenerate(golden):
    g = golden("voxelize_degenerate")
    out = ov.voxelize(g["far_points"], "Car")
    assert out["coordinate_buffer"].shape == (0, 3) and int(g["far_K"]) == 0
    assert out["feature_buffer"].shape == (0, 35, 7)
    one = ov.voxelize(g["one_points"], "Car")
    assert np.array_equal(one["feature_buffer"], g["one_feature"])
    assert np.array_equal(one["coordinate_buffer"], g["one_coord"])
    assert np.array_equal(one["number_buffer"], g["one_number"])
    empty = ov.voxelize(np.zeros((0, 4), np.float32), "Car")
    assert empty["number_buffer"].shape == (0,)


@pytest.mark.parametrize("cfg_id,target", [(2, "Car"), (3, "Pedestrian")])
def test_full_size_digest(golden, cfg_id, target):
    """Full-size frame: regenerate the cloud from its seed, replay the reference's
    shuffle (np.random.seed(7); np.random.shuffle), compare SHA-256 digests."""
    g = golden("voxelize_full_digest")
    w = synth.WORKLOADS[cfg_id]
    cloud = synth.synth_cloud(target, w["k0"], synth.frame_seed(cfg_id, 0), w["mean_extra"], w["T"])
    assert sha(cloud) == str(g[f"cfg{cfg_id}_cloud_sha"])
    np.random.seed(7)
    np.random.shuffle(cloud)
    assert sha(cloud) == str(g[f"cfg{cfg_id}_shuffled_sha"])
    out = ov.voxelize(cloud, target)
    assert out["coordinate_buffer"].shape[0] == int(g[f"cfg{cfg_id}_K"])
    assert sha(out["coordinate_buffer"]) == str(g[f"cfg{cfg_id}_coord_sha"])
    assert sha(out["number_buffer"]) == str(g[f"cfg{cfg_id}_number_sha"])
    assert sha(out["feature_buffer"]) == str(g[f"cfg{cfg_id}_feature_sha"])


def test_prepare_voxel_format():
    """dataset.py:101-119: (K,3) -> (K,4) with the sample index in column 0."""
    a = ov.voxelize(synth.synth_cloud("Car", 50, 1), "Car")
    b = ov.voxelize(synth.synth_cloud("Car", 60, 2), "Car")
    f, n, c = ov.prepare_voxel([a, b])
    assert c[0].shape == (a["coordinate_buffer"].shape[0], 4) and c[1].dtype == np.int64
    assert (c[0][:, 0] == 0).all() and (c[1][:, 0] == 1).all()
    assert np.array_equal(c[1][:, 1:], b["coordinate_buffer"])
    assert f[0] is a["feature_buffer"] and n[1] is b["number_buffer"]


def test_sorted_unique_property():
    """Size-independent properties: rows strictly ascending in linear key; counts
    in [1,T]; real slots carry the input points, padded slots are (0,0,0,0,-c)."""
    cloud = synth.synth_cloud("Car", 3000, 5)
    out = ov.voxelize(cloud, "Car")
    c = out["coordinate_buffer"]
    lin = (c[:, 0] * 400 + c[:, 1]) * 352 + c[:, 2]
    assert (np.diff(lin) > 0).all()
    n = out["number_buffer"]
    assert n.min() >= 1 and n.max() <= 35
    f = out["feature_buffer"]
    pad = np.arange(35)[None, :] >= n[:, None]
    assert (f[pad][:, :4] == 0).all()
    assert np.allclose(f[:, :, 4:7][pad].reshape(-1, 3),
                       -np.repeat((f[:, :, :3].sum(1) / n[:, None]), 35 - n, axis=0), atol=1e-5)
