"""Input pipeline of the train loop with the voxelizer on the device — the reference's `KITTIDataset`
(voxelnet/dataset.py:26-67) and `collate_fn` / `prepare_voxel` (dataset.py:70-119) re-cut for the GPU:

  reference:  DataLoader worker: read .bin -> pcl_to_voxels on the CPU (0.2-0.8 s, utils.py:10-100) -> (K,T,7) buffers
              through the worker pipe -> collate -> `.to(device)` of 6-14 MB per sample (model.py:302-303)
  here:       DataLoader worker: read .bin / label / image only -> main process: shuffle (utils.py:35), ONE pinned
              host->device copy of the raw (N,4) cloud (0.3-2 MB) -> `vn_voxelize_index/gather` on a side stream,
              no host synchronisation -> the reference's 7-tuple with the voxel buffers already in HBM.

`DeviceBatcher` wraps any iterable of per-sample 5-tuples lists (a DataLoader with `collate_fn=list`) and keeps one
batch in flight: batch i+1 is copied and voxelized while batch i trains.  Data augmentation (dataset.py:122-…,
`pcl_augmentation`) is outside the hot path and not provided; pass `augment=False`."""
import glob
import os

import numpy as np
import torch

from . import _lib
from .config import grid_config
from .voxelize import VoxelBatch, pipeline_stream, voxelize_device_async


def _read_image(path):
    """cv2.imread equivalent (dataset.py:50): HxWx3 uint8, BGR channel order; None when no decoder is installed"""
    try:
        from PIL import Image
    except ImportError:
        return None
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


class KITTIDataset(torch.utils.data.Dataset):
    """dataset.py:26-67 without the CPU voxelization: __getitem__ -> (tag, img, pcl (N,4) float32, labels, None)."""

    def __init__(self, data_dir, shuffle=True, augment=False, test=False, load_images=True):
        if augment:
            raise NotImplementedError("pcl_augmentation (dataset.py:122) is outside the accelerated path; use augment=False")
        self.data_dir, self.shuffle, self.test, self.load_images = data_dir, shuffle, test, load_images
        self.images = sorted(glob.glob(os.path.join(data_dir, "image_2") + "/*.png"))
        self.pcls = sorted(glob.glob(os.path.join(data_dir, "velodyne") + "/*.bin"))
        self.labels = sorted(glob.glob(os.path.join(data_dir, "label_2") + "/*.txt"))
        assert len(self.images) == len(self.pcls) == len(self.labels)          # dataset.py:40
        self.indices = list(range(len(self.images)))
        if self.shuffle:
            np.random.shuffle(self.indices)                                    # dataset.py:43-44

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        index = self.indices[idx]
        tag = os.path.split(self.images[index])[1][:-4]                        # dataset.py:48
        img = _read_image(self.images[index]) if self.load_images else None
        pcl = np.fromfile(self.pcls[index], dtype=np.float32).reshape(-1, 4)   # dataset.py:51
        labels = [] if self.test else [line for line in open(self.labels[index], "r").readlines()]
        return tag, img, pcl, labels, None


class DeviceCollate:
    """collate_fn (dataset.py:70-97) with pcl_to_voxels (utils.py:10-100) run on the device for the whole batch:
    parts = [(tag, img, pcl, labels, _)] -> the reference's 7-tuple, with x[2] / x[3] / x[4] = lists of DEVICE tensors
    feature (K_i,T,7) f32, number (K_i,) i64, coordinate (K_i,4) i64 [b,z,y,x].  Must run in the process that owns the
    GPU (not in a DataLoader worker)."""

    def __init__(self, device="cuda:0", target="Car", shuffle_points=True, fov_calib_dir=None, image_shape=(375, 1242)):
        """fov_calib_dir: RAW sweeps — crop every cloud to the camera field of view on the device before it is voxelized
        (the reference does this offline, preprocess_data.py:42-154, and trains on the rewritten .bin files):
        `<fov_calib_dir>/<tag>.txt` is the sample's KITTI object calibration file, the image size is the sample's image's
        (or image_shape when images are not loaded).  The crop keeps the input order, so shuffling the raw cloud first
        still hands the voxelizer a uniformly shuffled cropped cloud."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.VoxelnetHipError("DeviceCollate needs a HIP device (no CPU path)")
        self.grid = grid_config("Car" if target == "Car" else "Pedestrian")    # utils.py:24-33 ('Car' else ped/cyc)
        self.shuffle_points = shuffle_points
        self.fov_calib_dir, self.image_shape = fov_calib_dir, tuple(image_shape)
        self.stream = pipeline_stream(self.device)      # (shared with the target generator: see voxelize.pipeline_stream)

    def launch(self, parts):
        """enqueue the copies and the voxelization of one batch on the pipeline's stream; returns a handle"""
        handles = []
        with torch.cuda.stream(self.stream):
            for b, p in enumerate(parts):
                pcl = p[2]
                if self.shuffle_points:
                    np.random.shuffle(pcl)                                     # utils.py:35, in place like the reference
                host = torch.from_numpy(np.ascontiguousarray(pcl[:, :4], dtype=np.float32)).pin_memory()
                pts = host.to(self.device, non_blocking=True)
                if self.fov_calib_dir is not None:
                    from .fov import fov_crop_device, load_calib
                    P, Tr, R = load_calib(os.path.join(self.fov_calib_dir, str(p[0]) + ".txt"))
                    rows, cols = p[1].shape[:2] if p[1] is not None else self.image_shape
                    # padded form: no 4-byte read-back per sample (it would stall the host behind everything queued
                    # on this stream); the rows past the device-side count are NaN points, which the voxelizer drops
                    pts, _ = fov_crop_device(pts, P, Tr, R, rows, cols, padded=True)
                handles.append((voxelize_device_async(pts, self.grid, b, coord_cols=4), pts, host))
        return parts, handles

    def concat(self, launched):
        """second stage, still on the pipeline's stream: slice the capacity-sized outputs to K and make the concatenations
        the model starts with (RPN3D.detect) — off the train step's dependency chain.  `DeviceBatcher` calls this for
        batch i BEFORE it launches batch i+1, so the concatenation of batch i is queued behind batch i's own voxelizer
        only (queued behind the next batch's copies, crop and voxelization it would make step i wait for all of them)."""
        if len(launched) == 5:
            return launched
        parts, handles = launched
        feats, nums, coords = [], [], []
        for h, _, _ in handles:
            f, c, n = h.result()                    # waits for the 4-byte K copy of this sample only
            feats.append(f)
            coords.append(c)
            nums.append(n)
        if handles:
            feats = VoxelBatch.ahead(feats, self.stream, torch.float32)
            coords = VoxelBatch.ahead(coords, self.stream, torch.int64)
        return parts, handles, feats, nums, coords

    def finish(self, launched):
        parts, handles, feats, nums, coords = self.concat(launched)
        if handles:
            torch.cuda.current_stream().wait_event(handles[-1][0].event)       # consumer stream after the voxelizer
            for t in list(feats) + list(coords) + nums:
                t.record_stream(torch.cuda.current_stream())
        return ([p[0] for p in parts], np.array([p[3] for p in parts] + [None], dtype=object)[:-1], feats, nums, coords,
                np.array([p[1] for p in parts] + [None], dtype=object)[:-1],
                np.array([p[2] for p in parts] + [None], dtype=object)[:-1])

    def __call__(self, parts):
        return self.finish(self.launch(parts))


class DeviceBatcher:
    """Iterate an iterable of `parts` lists (e.g. DataLoader(ds, batch_size, collate_fn=list, num_workers=8)) as
    device-resident 7-tuples, one batch ahead: while the model trains on batch i, batch i+1 is being copied and
    voxelized on the pipeline's own stream."""

    def __init__(self, loader, device="cuda:0", target="Car", shuffle_points=True, fov_calib_dir=None, image_shape=(375, 1242)):
        self.loader = loader
        self.collate = DeviceCollate(device, target, shuffle_points, fov_calib_dir, image_shape)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        pending = None
        for parts in it:
            if pending is not None:
                pending = self.collate.concat(pending)     # batch i's concatenation in front of batch i+1's pipeline work
            launched = self.collate.launch(parts)
            if pending is not None:
                yield self.collate.finish(pending)
            pending = launched
        if pending is not None:
            yield self.collate.finish(pending)
