"""Host-side mirror of the reference's voxelization interface, backed by the HIP
voxelizer (csrc/voxelize.hip) through the C ABI.

  pcl_to_voxels   /root/reference/voxelnet/utils.py:10-100
  prepare_voxel   /root/reference/voxelnet/dataset.py:101-119
  collate_fn      /root/reference/voxelnet/dataset.py:70-97

Same names, argument meaning, return formats and side effects (pcl_to_voxels
shuffles its argument in place, utils.py:35).  `voxelize_device` is the
device-resident variant the train loop uses to skip the numpy round trip.
"""
import ctypes
import os

import numpy as np
import torch
import torch.utils.data

from . import _lib

_CAT_AHEAD = os.environ.get("VN_CAT_AHEAD") != "0"      # ("0": A/B aid, read once)
from .config import grid_config


def _grid_struct(g):
    return _lib.VnGrid(g.D, g.H, g.W, g.vz, g.vy, g.vx, g.ox, g.oy, g.oz, g.T)


def _stream():
    return _lib.raw_stream()


def voxelize_device(points, grid, batch_index=0, coord_cols=4):
    """points: (N,4) float32 CUDA tensor in processing order.
    Returns (feature (K,T,7) f32, coord (K,coord_cols) i64, number (K,) i64) on device.
    One 4-byte D2H read of K sizes the outputs (K is part of the reference's format)."""
    if not (points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 4):
        raise ValueError("points must be a CUDA float32 (N,4) tensor")
    points = points.contiguous()
    n = points.shape[0]
    dev = points.device
    gs = _grid_struct(grid)
    lib = _lib.load()
    ws_bytes = lib.vn_voxelize_workspace_bytes(n, ctypes.byref(gs))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    k_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    with _lib.on_device(dev):
        st = _stream()
        _lib.call("vn_voxelize_index", points.data_ptr(), n, ctypes.byref(gs), ws.data_ptr(), ws_bytes,
                  k_dev.data_ptr(), st)
        K = int(k_dev.item())
        feature = torch.empty((K, grid.T, 7), dtype=torch.float32, device=dev)
        coord = torch.empty((K, coord_cols), dtype=torch.int64, device=dev)
        number = torch.empty((K,), dtype=torch.int64, device=dev)
        _lib.call("vn_voxelize_gather", points.data_ptr(), n, ctypes.byref(gs), ws.data_ptr(), ws_bytes, K,
                  int(batch_index), coord_cols, feature.data_ptr(), coord.data_ptr(), number.data_ptr(), None, st)
    return feature, coord, number


class AsyncVoxels:
    """Handle of a voxelization that was enqueued WITHOUT a host read-back: the outputs are allocated at
    capacity (K <= N), the gather kernel takes K from device memory, and K travels to pinned host memory
    with an asynchronous copy.  result() waits for that copy only (normally long finished) and returns the
    reference-format views (K,T,7) / (K,cols) / (K,)."""

    def __init__(self, feature, coord, number, k_host, event):
        self._f, self._c, self._n, self._k, self._ev = feature, coord, number, k_host, event

    def result(self):
        if not self._ev.query():          # (normally long finished: a query costs ~1 us, a synchronize ~35)
            self._ev.synchronize()
        K = int(self._k[0])
        return self._f[:K], self._c[:K], self._n[:K]

    @property
    def event(self):
        return self._ev


class VoxelBuffers:
    """Reusable capacity-sized output/workspace buffers of one voxelization (input-pipeline slot)."""

    def __init__(self, n_points, grid, coord_cols, device):
        gs = _grid_struct(grid)
        self.n, self.cols = n_points, coord_cols
        self.cap = min(n_points, grid.cells)
        self.ws_bytes = _lib.load().vn_voxelize_workspace_bytes(n_points, ctypes.byref(gs))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.k_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.feature = torch.empty((self.cap, grid.T, 7), dtype=torch.float32, device=device)
        self.coord = torch.empty((self.cap, coord_cols), dtype=torch.int64, device=device)
        self.number = torch.empty((self.cap,), dtype=torch.int64, device=device)
        self.k_host = torch.empty(1, dtype=torch.int32, pin_memory=True)


def voxelize_device_async(points, grid, batch_index=0, coord_cols=4, buffers=None):
    """as voxelize_device, on the current stream, without any host synchronisation.
    buffers: a VoxelBuffers slot to write into (reused across steps by an input pipeline)."""
    if not (points.is_cuda and points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] == 4):
        raise ValueError("points must be a CUDA float32 (N,4) tensor")
    points = points.contiguous()
    n = points.shape[0]
    dev = points.device
    gs = _grid_struct(grid)
    lib = _lib.load()
    ws_bytes = lib.vn_voxelize_workspace_bytes(n, ctypes.byref(gs))
    cap = min(n, grid.cells)
    with _lib.on_device(dev):
        if buffers is not None:
            if buffers.n != n or buffers.cols != coord_cols:
                raise ValueError("VoxelBuffers slot was sized for a different cloud")
            ws, k_dev, feature, coord, number = buffers.ws, buffers.k_dev, buffers.feature, buffers.coord, buffers.number
        else:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            k_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            feature = torch.empty((cap, grid.T, 7), dtype=torch.float32, device=dev)
            coord = torch.empty((cap, coord_cols), dtype=torch.int64, device=dev)
            number = torch.empty((cap,), dtype=torch.int64, device=dev)
        st = _stream()
        from . import engine as E
        with E.section("voxelize", 16.0 * n):      # + 28*K*T + 40*K written, added by the caller once K is known
            _lib.call("vn_voxelize_index", points.data_ptr(), n, ctypes.byref(gs), ws.data_ptr(), ws_bytes,
                      k_dev.data_ptr(), st)
            _lib.call("vn_voxelize_gather", points.data_ptr(), n, ctypes.byref(gs), ws.data_ptr(), ws_bytes, cap,
                      int(batch_index), coord_cols, feature.data_ptr(), coord.data_ptr(), number.data_ptr(),
                      k_dev.data_ptr(), st)
        k_host = buffers.k_host if buffers is not None else torch.empty(1, dtype=torch.int32, pin_memory=True)
        k_host.copy_(k_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    return AsyncVoxels(feature, coord, number, k_host, ev)


_PIPELINE_STREAMS = {}


def pipeline_stream(device):
    """THE input-pipeline stream of a device: raw-cloud copies, field-of-view crop, voxelization, the batch concatenations and
    (RPN3D.forward) the target generation all queue here.  One stream for all of them, process-wide: the HIP runtime maps
    streams onto 4 hardware queues; with the training stream, the executor's side stream and (data-parallel runs) the
    reducer's communication stream that makes four — a fifth stream shares a queue with one of them and serialises against
    it (round 4: 540 instead of 556 point-clouds/s with the reducer attached), and MORE hardware queues are no way out
    (GPU_MAX_HW_QUEUES=8 with five busy streams: 248 point-clouds/s — the queues are time-sliced)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _PIPELINE_STREAMS.get(dev)
    if st is None:
        st = _PIPELINE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


class VoxelBatch(list):
    """The per-sample tensors of one batch — what collate_fn (dataset.py:80-96) puts into x[2] / x[4] — plus their
    concatenation, made AHEAD of the train step on the input pipeline's stream (`cat`, `cat_event`).  The model
    concatenates the lists first thing (model.py:93 via torch.cat in the reference's collate; RPN3D.detect here): two copy
    launches at the very start of the step's dependency chain, ~17 us, that the voxelizer's stream can do one step early.
    A plain list still works everywhere; this one only carries the ready-made result along."""
    cat = None
    cat_event = None

    @classmethod
    def ahead(cls, tensors, stream, dtype):
        out = cls(tensors)
        if len(out) > 1 and out[0].is_cuda and _CAT_AHEAD:   # ("0": A/B aid, the model concatenates)
            with torch.cuda.stream(stream):
                out.cat = torch.cat(list(out), dim=0).contiguous().to(dtype)
                out.cat_event = torch.cuda.Event()
                out.cat_event.record(stream)
        return out

    def take(self):
        """the concatenation, ordered into the current stream (or None: the caller concatenates)"""
        if self.cat is None:
            return None
        cur = _lib.current_stream(self.cat.device)
        cur.wait_event(self.cat_event)
        self.cat.record_stream(cur)
        return self.cat


def voxelize_host(points, grid, batch_index=0, coord_cols=3):
    """points: (N,4) float32 numpy array in processing order -> (feature (K,T,7) f32, coord (K,coord_cols) i64, number
    (K,) i64) numpy arrays, through the library's HOST entry points (vn_voxelize_host_index / _gather): no GPU is
    touched, so this is what runs inside forked DataLoader workers (dataset.py:58)."""
    pts = np.ascontiguousarray(points[:, :4], dtype=np.float32)
    n = pts.shape[0]
    gs = _grid_struct(grid)
    lib = _lib.load()
    ws_bytes = lib.vn_voxelize_host_workspace_bytes(n, ctypes.byref(gs))
    if ws_bytes == 0:
        raise _lib.VoxelnetHipError("vn_voxelize_host_workspace_bytes: unsupported grid / point count")
    ws = np.empty(ws_bytes, dtype=np.uint8)
    k = ctypes.c_int64(0)
    _lib.call("vn_voxelize_host_index", pts.ctypes.data, n, ctypes.byref(gs), ws.ctypes.data, ws_bytes, ctypes.byref(k))
    K = int(k.value)
    feature = np.empty((K, grid.T, 7), dtype=np.float32)
    coord = np.empty((K, coord_cols), dtype=np.int64)
    number = np.empty((K,), dtype=np.int64)
    _lib.call("vn_voxelize_host_gather", pts.ctypes.data, n, ctypes.byref(gs), ws.ctypes.data, ws_bytes, K, int(batch_index),
              coord_cols, feature.ctypes.data, coord.ctypes.data, number.ctypes.data)
    return feature, coord, number


def pcl_to_voxels(pcl, target, verbose=False, device=None):
    """Drop-in for utils.pcl_to_voxels (utils.py:10-100): numpy in, dict of numpy out.
    device: "cuda[:i]" -> the HIP voxelizer; "cpu" -> the library's host entry (vn_voxelize_host_*).  None (default, the
    reference's signature): the host entry inside a DataLoader worker process (where the reference calls it,
    dataset.py:58: a forked worker must not touch the GPU), the HIP voxelizer on cuda:0 otherwise.  Both give the same
    bits."""
    grid = grid_config("Car" if target == "Car" else "Pedestrian")   # utils.py:24-33: 'Car' else ped/cyc
    np.random.shuffle(pcl)                                           # utils.py:35, in place
    if device is None:
        device = "cpu" if torch.utils.data.get_worker_info() is not None else "cuda:0"
    if str(device) == "cpu":
        feature, coord, number = voxelize_host(pcl, grid, 0, coord_cols=3)
        voxel_dict = {"feature_buffer": feature, "coordinate_buffer": coord, "number_buffer": number}
    else:
        pts = torch.from_numpy(np.ascontiguousarray(pcl[:, :4], dtype=np.float32)).to(device)
        feature, coord, number = voxelize_device(pts, grid, 0, coord_cols=3)
        voxel_dict = {
            "feature_buffer": feature.cpu().numpy(),
            "coordinate_buffer": coord.cpu().numpy(),
            "number_buffer": number.cpu().numpy(),
        }
    if verbose:
        print(f"Coordinate buffer shape: {voxel_dict['coordinate_buffer'].shape}")
        print(f"Feature buffer shape: {voxel_dict['feature_buffer'].shape}")
        print(f"Number buffer shape: {voxel_dict['number_buffer'].shape}")
    return voxel_dict


def prepare_voxel(voxels):
    """dataset.py:101-119: list of voxel dicts -> (features, numbers, coordinates (K,4))."""
    features, numbers, coordinates = [], [], []
    for i, voxel in enumerate(voxels):
        features.append(voxel["feature_buffer"])
        numbers.append(voxel["number_buffer"])
        c = voxel["coordinate_buffer"]
        out = np.empty((c.shape[0], 4), dtype=c.dtype)
        out[:, 0] = i
        out[:, 1:] = c
        coordinates.append(out)
    return features, numbers, coordinates


def collate_fn(parts):
    """dataset.py:70-97: a list of dataset items (tag, rgb, raw_lidar, label, voxel dict) -> the reference's batch 7-tuple
    (tags, labels, [feature tensors], numbers, [coordinate tensors (K,4)], rgb, raw_lidar); the per-sample host arrays
    travel as object arrays, the voxel buffers as torch tensors."""
    tags, rgbs, clouds, labels, voxels = (list(column) for column in zip(*parts))
    feats, counts, coords = prepare_voxel(voxels)

    def ragged(seq):
        return np.array(seq, dtype=object)
    return (tags, ragged(labels), [torch.from_numpy(f) for f in feats], ragged(counts),
            [torch.from_numpy(c) for c in coords], ragged(rgbs), ragged(clouds))
