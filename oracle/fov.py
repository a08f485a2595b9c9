"""CPU restatement of the reference's camera field-of-view crop (voxelnet/preprocess_data.py) — TEST INFRASTRUCTURE
(the checker of vn_fov_crop; only tests/ import it).  Pinned by tests/golden/fov_crop.npz, which tools/gen_golden.py fov
makes by calling the imported reference's align_img_and_velo on a bundled KITTI frame.

NumPy float32 throughout, the same calls in the same order as the reference, so that the BLAS kernels that round the
reference's projections round these:
  prepare_velo_points   preprocess_data.py:42-49   reflectance > 0, homogeneous coordinate
  project_velo_to_img   preprocess_data.py:52-59   R_rect.dot(T.dot(p)), z >= 0, P.dot, divide by the third row
  align_img_and_velo    preprocess_data.py:62-103  col/row = int(np.round(.)), 0 < col < cols and 0 < row < rows
  main                  preprocess_data.py:151-154 the surviving [x, y, z, reflectance] rows as float32"""
import numpy as np


def fov_crop(points, P, Tr_velo_to_cam, R_cam_to_rect, rows, cols):
    """points (N,4) float32; P (3,4), Tr_velo_to_cam (4,4), R_cam_to_rect (4,4) float32 as load_calib returns them
    (preprocess_data.py:18-39).  -> (kept points (N',4) float32, their input indices (N',) int64)"""
    points = np.asarray(points, dtype=np.float32)
    idxs = points[:, 3] > 0                                           # :45
    pts3d = points[idxs, :].copy()
    pts3d[:, 3] = 1                                                   # :47
    p = pts3d.transpose()                                             # (4, N1)
    reflect = points[idxs, 3]
    cam = R_cam_to_rect.dot(Tr_velo_to_cam.dot(p))                    # :54
    front = cam[2, :] >= 0                                            # :56
    pix = P.dot(cam[:, front])                                        # :57
    pix = pix / pix[2, :]                                             # :59
    p3 = p[:, front]
    reflect = reflect[front]
    src = np.flatnonzero(idxs)[front]
    with np.errstate(invalid="ignore"):
        col = np.round(pix[0, :])                                     # :83 (int(np.round(x)): round half to even)
        row = np.round(pix[1, :])                                     # :84
        keep = (col < cols) & (row < rows) & (row > 0) & (col > 0)    # :86
    out = np.stack([p3[0, keep], p3[1, keep], p3[2, keep], reflect[keep]], axis=1).astype(np.float32)
    return out, src[keep]
