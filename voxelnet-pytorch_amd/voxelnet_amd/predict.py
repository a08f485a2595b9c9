"""Inference tail of the reference's `RPN3D.predict` (voxelnet/model.py:364-395) on the device: box decoding
(utils.deltas_to_boxes_3d, utils.py:476-489), score filter, stand-up rectangles and NMS (model.filter_boxes,
model.py:28-57; utils.nms, utils.py:492-553) through `vn_rpn_predict` (csrc/predict.hip).  The probability and delta
maps never leave HBM; only the <= NMS_POST_TOPK kept boxes per sample come back.  No CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib
from .targets import CLASS_CFG, generate_anchors

SCORE_THRES, NMS_THRES, NMS_POST_TOPK = 0.96, 0.1, 20          # config.py:95-98 (cfg.RPN)


class BoxDecoder:
    def __init__(self, cls_name="Car", device="cuda:0", anchors=None):
        self.cls_name = cls_name
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.VoxelnetHipError("BoxDecoder needs a HIP device (no CPU path)")
        self.anchors = generate_anchors(cls_name) if anchors is None else np.asarray(anchors, dtype=np.float64)
        self._anchors_dev = torch.from_numpy(np.ascontiguousarray(self.anchors.reshape(-1, 7))).to(self.device)
        self.n_anchors = self._anchors_dev.shape[0]
        self.anchor_h = float(CLASS_CFG[cls_name]["h"])

    def __call__(self, probs, deltas, score_thres=SCORE_THRES, nms_thres=NMS_THRES, top_k=NMS_POST_TOPK):
        """probs (B,2,h,w), deltas (B,14,h,w) fp32 device tensors -> ([boxes (n_i,7) f32 numpy], [scores (n_i,) f32 numpy])"""
        if not (probs.is_cuda and deltas.is_cuda):
            raise _lib.VoxelnetHipError("predict: probs / deltas must be HIP tensors (no CPU path)")
        probs, deltas = probs.detach().float().contiguous(), deltas.detach().float().contiguous()
        B, N = probs.shape[0], self.n_anchors
        if probs[0].numel() != N or deltas[0].numel() != 7 * N:
            raise ValueError(f"maps of {probs[0].numel()} / {deltas[0].numel()} elements do not match {N} anchors")
        dev = probs.device
        boxes = torch.zeros((B, top_k, 7), dtype=torch.float32, device=dev)
        scores = torch.zeros((B, top_k), dtype=torch.float32, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        nbytes = _lib.load().vn_rpn_predict_workspace_bytes(B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            _lib.call("vn_rpn_predict", probs.data_ptr(), deltas.data_ptr(), self._anchors_dev.data_ptr(), B, N,
                      float(score_thres), float(nms_thres), int(top_k), self.anchor_h, boxes.data_ptr(), scores.data_ptr(),
                      counts.data_ptr(), ws.data_ptr(), nbytes, _lib.raw_stream())
        cnt = counts.cpu().numpy()
        bh, sh = boxes.cpu().numpy(), scores.cpu().numpy()
        return [bh[b, :cnt[b]].copy() for b in range(B)], [sh[b, :cnt[b]].copy() for b in range(B)]
