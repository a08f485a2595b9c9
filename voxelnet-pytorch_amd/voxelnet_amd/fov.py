"""Camera field-of-view crop of a raw Velodyne sweep on the device — the reference's offline preprocessing
(voxelnet/preprocess_data.py:42-103; `main()` rewrites every velodyne .bin with the surviving rows, :151-154) as a stage
of the input pipeline (csrc/fov.hip, `vn_fov_crop`): the raw (N,4) cloud goes to the device once, is cropped there and
feeds the voxelizer.  No CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib


def load_calib(calib_path):
    """preprocess_data.py:18-39: KITTI object calibration file -> (P2 (3,4), Tr_velo_to_cam (4,4), R_cam_to_rect (4,4))
    float32"""
    with open(calib_path) as fh:
        lines = [line.split()[1:] for line in fh.readlines()][:-1]
    P = np.array(lines[2], dtype=np.float64).reshape(3, 4)
    Tr = np.concatenate([np.array(lines[5], dtype=np.float64).reshape(3, 4), np.array([[0.0, 0.0, 0.0, 1.0]])], 0)
    R = np.eye(4)
    R[:3, :3] = np.array(lines[4][:9], dtype=np.float64).reshape(3, 3)
    return P.astype(np.float32), Tr.astype(np.float32), R.astype(np.float32)


def fov_crop_device(points, P, Tr_velo_to_cam, R_cam_to_rect, image_rows, image_cols, return_index=False, padded=False):
    """points: (N,4) float32 CUDA tensor [x,y,z,reflectance].  -> the surviving rows in input order (a view of a
    capacity-sized buffer, sliced to the count: one device->host read of 4 bytes on the current stream)
    [, their input row numbers (int32)].
    padded=True: no host synchronisation — returns (the whole capacity-sized (N,4) buffer whose rows past the count are
    NaN points, the count as a device int32 tensor); the voxelizer drops NaN points like any out-of-range point, so
    voxelizing the padded buffer equals voxelizing the cropped cloud bit for bit (the input pipeline's form)."""
    if not (torch.is_tensor(points) and points.is_cuda and points.dtype == torch.float32 and points.dim() == 2
            and points.shape[1] == 4):
        raise _lib.VoxelnetHipError("fov_crop_device needs an (N,4) float32 HIP tensor (there is no CPU path)")
    points = points.contiguous()
    n = points.shape[0]
    mats = [np.ascontiguousarray(m, dtype=np.float32) for m in (P, Tr_velo_to_cam, R_cam_to_rect)]
    if mats[0].shape != (3, 4) or mats[1].shape != (4, 4) or mats[2].shape != (4, 4):
        raise ValueError("calibration: P (3,4), Tr_velo_to_cam (4,4), R_cam_to_rect (4,4)")
    dev = points.device
    with _lib.on_device(dev):
        out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=dev)
        index = torch.empty(max(n, 1), dtype=torch.int32, device=dev) if return_index else None
        count = torch.empty(1, dtype=torch.int32, device=dev)
        nbytes = _lib.load().vn_fov_crop_workspace_bytes(n)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _lib.call("vn_fov_crop", points.data_ptr(), n, mats[0].ctypes.data, mats[1].ctypes.data, mats[2].ctypes.data,
                  int(image_rows), int(image_cols), out.data_ptr(), index.data_ptr() if index is not None else None,
                  count.data_ptr(), ws.data_ptr(), ws.numel(), _lib.raw_stream())
        if padded:
            return (out[:n], count, index) if return_index else (out[:n], count)
        k = int(count.item())
    return (out[:k], index[:k]) if return_index else out[:k]
