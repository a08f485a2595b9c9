"""GPU: vn_fov_crop (csrc/fov.hip through voxelnet_amd.fov) against tests/golden/fov_crop.npz — the imported reference's
align_img_and_velo (preprocess_data.py:62-103) on a bundled KITTI frame plus points on the decision boundaries — and
against the oracle on a full-size synthetic sweep: the surviving rows bit-equal and in input order (index set exact)."""
import numpy as np
import pytest
import torch

from oracle import fov as of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fov_crop_matches_the_reference_fixture(golden):
    from voxelnet_amd.fov import fov_crop_device
    g = golden("fov_crop")
    rows, cols = (int(v) for v in g["image_shape"])
    out, idx = fov_crop_device(torch.from_numpy(g["points"]).to(DEV), g["P"], g["Tr"], g["R"], rows, cols, return_index=True)
    assert np.array_equal(out.cpu().numpy(), g["kept"])
    assert np.array_equal(g["points"][idx.cpu().numpy()], g["kept"])


def test_fov_crop_full_sweep_and_edges(golden):
    from voxelnet_amd.fov import fov_crop_device
    g = golden("fov_crop")
    rows, cols = (int(v) for v in g["image_shape"])
    rng = np.random.default_rng(5)
    n = 121_015                                                      # a raw KITTI sweep
    pts = np.stack([rng.uniform(-80, 80, n), rng.uniform(-80, 80, n), rng.uniform(-3, 2, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    pts[rng.integers(0, n, 2000), 3] = 0.0
    ref, ridx = of.fov_crop(pts, g["P"], g["Tr"], g["R"], rows, cols)
    out, idx = fov_crop_device(torch.from_numpy(pts).to(DEV), g["P"], g["Tr"], g["R"], rows, cols, return_index=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx) and np.array_equal(out.cpu().numpy(), ref)
    assert 0.05 * n < out.shape[0] < 0.5 * n
    # empty input, a single point, nothing kept
    for sub in (pts[:0], pts[ridx[:1]], pts[pts[:, 0] < -1][:1000]):
        o = fov_crop_device(torch.from_numpy(np.ascontiguousarray(sub)).to(DEV), g["P"], g["Tr"], g["R"], rows, cols)
        r, _ = of.fov_crop(sub, g["P"], g["Tr"], g["R"], rows, cols)
        assert np.array_equal(o.cpu().numpy(), r)
    with pytest.raises(Exception):
        fov_crop_device(torch.from_numpy(pts), g["P"], g["Tr"], g["R"], rows, cols)         # CPU tensor: no CPU path


def test_device_batcher_crops_raw_sweeps(golden, tmp_path):
    """DeviceBatcher(fov_calib_dir=...): raw sweep -> (shuffle on the host) -> device crop -> device voxelizer gives the
    voxel buffers of the oracle voxelizer run on the oracle-cropped cloud in the same order."""
    import os
    from oracle import voxelize as ov
    from voxelnet_amd import dataset as D
    g = golden("fov_crop")
    rows, cols = (int(v) for v in g["image_shape"])
    root = str(tmp_path / "kitti")
    for sub in ("image_2", "velodyne", "label_2", "calib"):
        os.makedirs(os.path.join(root, sub))
    rng = np.random.default_rng(9)
    for i in range(2):
        tag = f"{i:06d}"
        pts = np.stack([rng.uniform(0, 70, 30000), rng.uniform(-40, 40, 30000), rng.uniform(-3, 1, 30000), rng.uniform(0, 1, 30000)], 1)
        pts.astype(np.float32).tofile(os.path.join(root, "velodyne", tag + ".bin"))
        open(os.path.join(root, "image_2", tag + ".png"), "wb").close()
        open(os.path.join(root, "label_2", tag + ".txt"), "w").write("Car 0.00 0 0.00 0.00 0.00 0.00 0.00 1.50 1.60 3.90 1.00 1.50 20.00 0.10\n")
        with open(os.path.join(root, "calib", tag + ".txt"), "w") as fh:
            def fmt(name, a):
                return name + ": " + " ".join(f"{v:.12e}" for v in np.asarray(a).reshape(-1))
            fh.write("\n".join([fmt("P0", g["P"]), fmt("P1", g["P"]), fmt("P2", g["P"]), fmt("P3", g["P"]), fmt("R0_rect", g["R"][:3, :3]),
                                fmt("Tr_velo_to_cam", g["Tr"][:3]), fmt("Tr_imu_to_velo", g["Tr"][:3])]) + "\n")
    ds = D.KITTIDataset(root, shuffle=False, load_images=False)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=list, num_workers=0)
    np.random.seed(77)
    (batch,) = list(D.DeviceBatcher(loader, DEV, "Car", fov_calib_dir=os.path.join(root, "calib"), image_shape=(rows, cols)))
    np.random.seed(77)
    for i in range(2):
        cloud = np.fromfile(os.path.join(root, "velodyne", f"{i:06d}.bin"), dtype=np.float32).reshape(-1, 4)
        np.random.shuffle(cloud)
        from voxelnet_amd.fov import load_calib
        P, Tr, R = load_calib(os.path.join(root, "calib", f"{i:06d}.txt"))
        cropped, _ = of.fov_crop(cloud, P, Tr, R, rows, cols)
        assert 0 < cropped.shape[0] < cloud.shape[0]
        ref = ov.voxelize(cropped, "Car")
        assert np.array_equal(batch[2][i].cpu().numpy(), ref["feature_buffer"])
        assert np.array_equal(batch[4][i].cpu().numpy()[:, 1:], ref["coordinate_buffer"])


def test_padded_crop_feeds_the_voxelizer_without_a_read_back(golden):
    """fov_crop_device(padded=True): the capacity-sized buffer with NaN points past the device-side count (no host
    synchronisation) voxelizes to the same bits as the cropped cloud — what DeviceCollate hands the voxelizer."""
    from oracle import voxelize as ov
    from voxelnet_amd.config import grid_config
    from voxelnet_amd.fov import fov_crop_device
    from voxelnet_amd.voxelize import voxelize_device
    g = golden("fov_crop")
    rows, cols = (int(v) for v in g["image_shape"])
    rng = np.random.default_rng(11)
    n = 60_000
    pts = np.stack([rng.uniform(-10, 70, n), rng.uniform(-40, 40, n), rng.uniform(-3, 1, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    ref, _ = of.fov_crop(pts, g["P"], g["Tr"], g["R"], rows, cols)
    out, count = fov_crop_device(torch.from_numpy(pts).to(DEV), g["P"], g["Tr"], g["R"], rows, cols, padded=True)
    assert out.shape == (n, 4) and count.dtype == torch.int32 and count.is_cuda
    k = int(count.item())
    assert k == ref.shape[0] and 0 < k < n
    o = out.cpu().numpy()
    assert np.array_equal(o[:k], ref) and np.isnan(o[k:]).all()
    grid = grid_config("Car")
    f, c, m = voxelize_device(out, grid, 0, coord_cols=3)
    want = ov.voxelize(ref, "Car")
    assert np.array_equal(c.cpu().numpy(), want["coordinate_buffer"])
    assert np.array_equal(m.cpu().numpy(), want["number_buffer"])
    assert np.array_equal(f.cpu().numpy(), want["feature_buffer"])
