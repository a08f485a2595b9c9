"""RPN training targets — the reference's `generate_anchors` (voxelnet/utils.py:104-130) and `generate_targets`
(utils.py:376-473; called by RPN3D.forward, model.py:309) with the per-anchor work on the device (csrc/targets.hip,
`vn_rpn_targets`).

Host side (O(number of boxes) NumPy, as in the reference): KITTI label lines -> lidar boxes (utils.py:147-210), their
stand-up rectangles (utils.py:230-252, 283-330).  Device side: 70,400 anchors x boxes IoU with the reference's
conventions and quirks, the positive / negative assignment and the regression encoding.  Outputs stay on the device
in the layouts `RPN3D.loss` takes — no float64 host arrays, no six host-to-device copies per step (model.py:327-332).
There is no CPU fallback: the HIP library must be present."""
import ctypes
import os

import numpy as np
import torch

from . import _lib

MAX_GT = 128     # VN_TARGETS_MAX_GT (include/voxelnet_hip.h)

# voxelnet/config.py:36-92 (anchor geometry, IoU thresholds) and :99-111 (mean KITTI calibration)
CLASS_CFG = {
    "Car": dict(x=(0.0, 70.4), y=(-40.0, 40.0), fw=176, fh=200, l=3.9, w=1.6, h=1.56, z=-1.0 - 1.56 / 2, pos_iou=0.6,
                neg_iou=0.45, accept=("Car", "Van")),
    "Pedestrian": dict(x=(0.0, 48.0), y=(-20.0, 20.0), fw=120, fh=100, l=0.8, w=0.6, h=1.73, z=-0.6 - 1.73 / 2,
                       pos_iou=0.5, neg_iou=0.35, accept=("Pedestrian",)),
    "Cyclist": dict(x=(0.0, 48.0), y=(-20.0, 20.0), fw=120, fh=100, l=1.76, w=0.6, h=1.73, z=-0.6 - 1.73 / 2,
                    pos_iou=0.5, neg_iou=0.35, accept=("Cyclist",)),
}
_T_VELO_2_CAM = np.array([[7.49916597e-03, -9.99971248e-01, -8.65110297e-04, -6.71807577e-03],
                          [1.18652889e-02, 9.54520517e-04, -9.99910318e-01, -7.33152811e-02],
                          [9.99882833e-01, 7.49141178e-03, 1.18719929e-02, -2.78557062e-01],
                          [0, 0, 0, 1]])
_R_RECT_0 = np.array([[0.99992475, 0.00975976, -0.00734152, 0],
                      [-0.0097913, 0.99994262, -0.00430371, 0],
                      [0.00729911, 0.0043753, 0.99996319, 0],
                      [0, 0, 0, 1]])
_R_RECT_0_INV, _T_VELO_2_CAM_INV = np.linalg.inv(_R_RECT_0), np.linalg.inv(_T_VELO_2_CAM)   # (utils.py:168-169 inverts per call)


def generate_anchors(cls_name="Car"):
    """utils.py:104-130: (FEATURE_HEIGHT, FEATURE_WIDTH, 2, 7) float64 anchors [x, y, z, h, w, l, r], r in {0, pi/2}."""
    c = CLASS_CFG[cls_name]
    gx, gy = np.meshgrid(np.linspace(c["x"][0], c["x"][1], c["fw"]), np.linspace(c["y"][0], c["y"][1], c["fh"]))
    a = np.empty((c["fh"], c["fw"], 2, 7))
    a[..., 0] = gx[..., None]
    a[..., 1] = gy[..., None]
    a[..., 2], a[..., 3], a[..., 4], a[..., 5] = c["z"], c["h"], c["w"], c["l"]
    a[..., 0, 6] = 0
    a[..., 1, 6] = 90 / 180 * np.pi
    return a


def _limit_angle(angle):
    """utils.py:133-144"""
    while angle >= np.pi / 2:
        angle -= np.pi
    while angle < -np.pi / 2:
        angle += np.pi
    return np.pi / 2 if abs(angle + np.pi / 2) < 5 / 180 * np.pi else angle


def label_to_gt_box_3d(labels, cls_name="Car", coordinate="lidar"):
    """utils.py:178-210 (+ camera_to_lidar_box, :163-174): label lines of every sample -> list of (G_i, 7) float64
    boxes (x, y, z, h, w, l, r), in lidar coordinates by the mean calibration unless coordinate == 'camera'."""
    accept = CLASS_CFG[cls_name]["accept"] if cls_name in CLASS_CFG else ()
    r_inv, t_inv = _R_RECT_0_INV, _T_VELO_2_CAM_INV
    out = []
    for label in labels:
        rows = []
        for line in label:
            f = line.split()
            if accept and f[0] not in accept:
                continue
            h, w, l, x, y, z, ry = (float(v) for v in f[-7:])
            if coordinate == "lidar":
                p = np.matmul(t_inv, np.matmul(r_inv, np.array([x, y, z, 1])))
                rows.append([p[0], p[1], p[2], h, w, l, _limit_angle(-ry - np.pi / 2)])
            else:
                rows.append([x, y, z, h, w, l, ry])
        out.append(np.array(rows, dtype=np.float64).reshape(-1, 7))
    return out


def lidar_box_to_label_line(cls_name, box):
    """inverse of label_to_gt_box_3d for one box (x, y, z, h, w, l, r) in lidar coordinates: the KITTI label line (camera
    coordinates by the mean calibration, utils.py lidar_to_camera) the dataset would hand to RPN3D.forward — synthetic
    labels for bench.py / tests"""
    x, y, z, h, w, l, rz = (float(v) for v in box)
    p = np.matmul(_R_RECT_0, np.matmul(_T_VELO_2_CAM, np.array([x, y, z, 1.0])))
    return (f"{cls_name} 0.00 0 0.00 0.00 0.00 0.00 0.00 {h:.2f} {w:.2f} {l:.2f} {p[0]:.2f} {p[1]:.2f} {p[2]:.2f} "
            f"{-rz - np.pi / 2:.2f}")


def gt_standup_boxes(gt):
    """corner_to_standup_box2d(center_to_corner_box_2d(gt[:, [0,1,4,5,6]])) (utils.py:402-406): axis-aligned hull of
    the rotated footprint; float64 rotation, corners stored as float32 (utils.py:293), result float32 (utils.py:410)."""
    out = np.zeros((gt.shape[0], 4), dtype=np.float32)
    for i, (x, y, _, _, w, l, yaw) in enumerate(gt):
        foot = np.array([[-l / 2, -l / 2, l / 2, l / 2], [w / 2, -w / 2, -w / 2, w / 2], [0.0, 0.0, 0.0, 0.0]])
        rot = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
        c = (np.dot(rot, foot) + np.array([[x], [y], [0.0]])).T.astype(np.float32)
        out[i] = (c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max())
    return out


# host-side memo of label text -> boxes (VN_LABEL_CACHE=0: parse every time, as the reference does; the DEVICE target
# generation runs every step either way — only host work is remembered, and the step is GPU-bound: same point-clouds/s)
_LABEL_CACHE = os.environ.get("VN_LABEL_CACHE", "1") != "0"


class TargetGenerator:
    """Device-resident anchors of one class + the launch: `gen(labels)` -> (pos_equal_one (B,h,w,2), neg_equal_one
    (B,h,w,2), targets (B,h,w,14)) float32 tensors on `device`, the arrays of utils.generate_targets (utils.py:473)."""

    def __init__(self, cls_name="Car", device="cuda:0", anchors=None):
        self.cls_name = cls_name
        self.cfg = CLASS_CFG[cls_name]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.VoxelnetHipError("TargetGenerator needs a HIP device (no CPU path)")
        self.anchors = generate_anchors(cls_name) if anchors is None else np.asarray(anchors, dtype=np.float64)
        self.shape = self.anchors.shape[:2]
        self._anchors_dev = torch.from_numpy(np.ascontiguousarray(self.anchors.reshape(-1, 7))).to(self.device)
        self.n_anchors = self._anchors_dev.shape[0]
        self._parsed = {}          # (label lines, coordinate) -> (G, 7) float64 boxes
        self._standup = {}         # id-free cache of gt_standup_boxes, keyed by the boxes' bytes

    def from_boxes(self, gt_boxes):
        """gt_boxes: list of (G_i, 7) float64 lidar boxes per sample"""
        B = len(gt_boxes)
        G = max([b.shape[0] for b in gt_boxes] + [1])
        if G > MAX_GT:
            raise _lib.VoxelnetHipError(f"{G} ground-truth boxes in one sample; vn_rpn_targets takes at most {MAX_GT}")
        gt = np.zeros((B, G, 7), dtype=np.float64)
        g2 = np.zeros((B, G, 4), dtype=np.float32)
        cnt = np.zeros(B, dtype=np.int32)
        for b, boxes in enumerate(gt_boxes):
            n = boxes.shape[0]
            cnt[b] = n
            if n:
                gt[b, :n] = boxes
                key = boxes.tobytes()
                su = self._standup.get(key) if _LABEL_CACHE else None
                if su is None:
                    if len(self._standup) >= 8192:
                        self._standup.clear()
                    su = self._standup[key] = gt_standup_boxes(boxes)
                g2[b, :n] = su
        dev = self.device
        # pinned staging + asynchronous copies on the current stream: a blocking copy from pageable memory would make the
        # host wait for everything queued before it — every step — and the train loop's enqueue could no longer run ahead
        # of the GPU (the caching host allocator keeps a pinned block until the copy that reads it has run)
        gt_d, g2_d, cnt_d = (torch.from_numpy(a).pin_memory().to(dev, non_blocking=True) for a in (gt, g2, cnt))
        N = self.n_anchors
        pos = torch.empty((B, *self.shape, 2), dtype=torch.float32, device=dev)
        neg = torch.empty((B, *self.shape, 2), dtype=torch.float32, device=dev)
        tgt = torch.empty((B, *self.shape, 14), dtype=torch.float32, device=dev)
        nbytes = _lib.load().vn_rpn_targets_workspace_bytes(B, N, G)
        if nbytes == 0:
            raise _lib.VoxelnetHipError("vn_rpn_targets_workspace_bytes: unsupported sizes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            _lib.call("vn_rpn_targets", self._anchors_dev.data_ptr(), N, gt_d.data_ptr(), g2_d.data_ptr(), cnt_d.data_ptr(),
                      B, G, float(self.cfg["pos_iou"]), float(self.cfg["neg_iou"]), float(self.cfg["h"]), pos.data_ptr(),
                      neg.data_ptr(), tgt.data_ptr(), ws.data_ptr(), nbytes,
                      _lib.raw_stream())
        return pos, neg, tgt

    def __call__(self, labels, feature_map_shape=None, coordinate="lidar"):
        if feature_map_shape is not None and tuple(feature_map_shape) != tuple(self.shape):
            raise ValueError(f"feature map {tuple(feature_map_shape)} does not match the anchors {tuple(self.shape)}")
        # label lines -> lidar boxes is a pure function of the TEXT of a sample's label: remembered per sample (an epoch
        # visits every sample's label once per epoch; 0.25 ms of the host's 1.8 ms per step went into re-parsing, NumPy
        # 4-vectors one box at a time — tools/host_cprofile.py).  Keyed by the text itself, not by object identity.
        boxes = []
        for label in labels:
            key = (tuple(label), coordinate)
            hit = self._parsed.get(key) if _LABEL_CACHE else None
            if hit is None:
                if len(self._parsed) >= 8192:
                    self._parsed.clear()
                hit = label_to_gt_box_3d([label], self.cls_name, coordinate)[0]
                hit.setflags(write=False)
                self._parsed[key] = hit
            boxes.append(hit)
        return self.from_boxes(boxes)


def generate_targets(labels, feature_map_shape, anchors, cls_name="Car", coordinate="lidar", device="cuda:0"):
    """utils.py:376-382 signature; returns device tensors instead of float64 host arrays."""
    return TargetGenerator(cls_name, device, anchors)(labels, feature_map_shape, coordinate)
