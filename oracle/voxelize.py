"""ORACLE — test infrastructure, NOT product code.

ctypes front-end of oracle/voxelize_ref.c, the CPU restatement of
/root/reference/voxelnet/utils.py:10-100 (pcl_to_voxels, shuffle excluded) and
/root/reference/voxelnet/dataset.py:101-119 (prepare_voxel).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Pinned against the imported reference by tests/golden/voxelize_*.npz.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# utils.py:24-33 literals (z,y,x voxel size; x,y,z offsets)
GRIDS = {
    "Car": dict(D=10, H=400, W=352, vz=0.4, vy=0.2, vx=0.2, ox=0.0, oy=40.0, oz=3.0, T=35),
    "Pedestrian": dict(D=10, H=200, W=240, vz=0.4, vy=0.2, vx=0.2, ox=0.0, oy=20.0, oz=3.0, T=45),
    "Cyclist": dict(D=10, H=200, W=240, vz=0.4, vy=0.2, vx=0.2, ox=0.0, oy=20.0, oz=3.0, T=45),
}


class _Grid(ctypes.Structure):
    _fields_ = [("D", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("vz", ctypes.c_float), ("vy", ctypes.c_float), ("vx", ctypes.c_float),
                ("ox", ctypes.c_float), ("oy", ctypes.c_float), ("oz", ctypes.c_float),
                ("T", ctypes.c_int32)]


def build():
    """Compile liboracle.so (gcc).  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("VN_ORACLE_LIB")        # tests/test_sanitizers.py: the ASan build (oracle/Makefile `asan`)
        if not path:
            path = os.path.join(_HERE, "liboracle.so")
            if not os.path.exists(path):
                build()
        lib = ctypes.CDLL(path)
        lib.vn_oracle_voxelize.restype = ctypes.c_int64
        lib.vn_oracle_voxelize.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(_Grid),
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]
        lib.vn_oracle_pad_coords.restype = None
        lib.vn_oracle_pad_coords.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_void_p]
        _LIB = lib
    return _LIB


def grid_for(target, T=None, **override):
    g = dict(GRIDS["Car" if target == "Car" else "Pedestrian"])
    g.update(override)
    if T is not None:
        g["T"] = int(T)
    return g


def voxelize(pcl, target="Car", T=None, **override):
    """pcl: (N,4) float32 in processing order (already shuffled).
    Returns the reference's dict (utils.py:90-94)."""
    pcl = np.ascontiguousarray(pcl, dtype=np.float32)
    assert pcl.ndim == 2 and pcl.shape[1] == 4
    g = grid_for(target, T, **override)
    cg = _Grid(**g)
    n = pcl.shape[0]
    kcap = max(1, min(n, g["D"] * g["H"] * g["W"]))
    feat = np.empty((kcap, g["T"], 7), np.float32)
    coord = np.empty((kcap, 3), np.int64)
    number = np.empty((kcap,), np.int64)
    k = _lib().vn_oracle_voxelize(pcl.ctypes.data, n, ctypes.byref(cg), kcap,
                                  feat.ctypes.data, coord.ctypes.data, number.ctypes.data)
    if k < 0:
        raise RuntimeError(f"vn_oracle_voxelize failed: {k}")
    return {
        "feature_buffer": feat[:k].copy(),
        "coordinate_buffer": coord[:k].copy(),
        "number_buffer": number[:k].copy(),
    }


def prepare_voxel(voxels):
    """dataset.py:101-119: list of voxel dicts -> (features, numbers, coordinates(K,4))."""
    features, numbers, coordinates = [], [], []
    for i, v in enumerate(voxels):
        c3 = np.ascontiguousarray(v["coordinate_buffer"], dtype=np.int64)
        c4 = np.empty((c3.shape[0], 4), np.int64)
        _lib().vn_oracle_pad_coords(c3.ctypes.data, c3.shape[0], i, c4.ctypes.data)
        features.append(v["feature_buffer"])
        numbers.append(v["number_buffer"])
        coordinates.append(c4)
    return features, numbers, coordinates


def voxelize_numpy(pcl, target="Car", T=None, **override):
    """Vectorised numpy restatement of the same function (second opinion for the
    C version on small clouds; not used for timing)."""
    g = grid_for(target, T, **override)
    pcl = np.asarray(pcl, dtype=np.float32)
    off = np.array([g["ox"], g["oy"], g["oz"]], np.float32)
    vs = np.array([g["vz"], g["vy"], g["vx"]], np.float32)
    gs = np.array([g["D"], g["H"], g["W"]])
    idx = np.floor((pcl[:, :3] + off)[:, ::-1] / vs)
    ok = np.all((idx >= 0) & (idx < gs), axis=1)
    pcl, idx = pcl[ok], idx[ok].astype(np.int64)
    lin = (idx[:, 0] * g["H"] + idx[:, 1]) * g["W"] + idx[:, 2]
    uniq, inv = np.unique(lin, return_inverse=True)
    K, Tn = len(uniq), g["T"]
    order = np.argsort(inv, kind="stable")
    inv_sorted = inv[order]
    start = np.searchsorted(inv_sorted, np.arange(K))
    rank = np.arange(len(inv)) - start[inv_sorted]
    keep = rank < Tn
    feat = np.zeros((K, Tn, 7), np.float32)
    feat[inv_sorted[keep], rank[keep], :4] = pcl[order[keep]]
    number = np.minimum(np.bincount(inv, minlength=K), Tn).astype(np.int64)
    s = np.zeros((K, 3), np.float32)
    for t in range(Tn):  # sequential float32 accumulation, slot order
        s = s + feat[:, t, :3]
    cen = s.astype(np.float64) / number.reshape(K, 1).astype(np.float64)
    feat[:, :, 4:7] = (feat[:, :, :3].astype(np.float64) - cen[:, None, :]).astype(np.float32)
    coord = np.stack([uniq // (g["H"] * g["W"]), (uniq // g["W"]) % g["H"], uniq % g["W"]], 1)
    return {"feature_buffer": feat, "coordinate_buffer": coord.astype(np.int64),
            "number_buffer": number}
