"""ORACLE (test infrastructure, not the product path): NumPy restatement of the reference's inference tail —
`deltas_to_boxes_3d` (utils.py:476-489), `filter_boxes` (model.py:28-57) with `nms` (utils.py:492-553), and the
box/score part of `RPN3D.predict` (model.py:364-395).  Pinned by tests/golden/predict_car.npz (tools/gen_golden.py
`predict`, from the imported reference with CPU tensors).

Reference behaviour kept on purpose:
  * `deltas.reshape(B, -1, 7)` and `probs.reshape(B, -1)` act on the NCHW arrays WITHOUT a permute (utils.py:478,
    model.py:384): "box j" is built from the flat elements 7j..7j+6 of the (14,h,w) delta map and scored by flat element
    j of the (2,h,w) probability map, while anchor j is (iy, ix, rotation) — mirrored as is;
  * boxes are float32 (np.zeros_like(deltas)) holding float64 results; exp is evaluated in float32;
  * only the NMS_POST_TOPK = 20 highest-scoring boxes above SCORE_THRES enter the NMS at all (utils.py:510), the
    stand-up rectangles there are float64 and areas have no "+1";
  * ties in score: the reference sorts with torch.sort(stable=False) (utils.py:509), which leaves the order of equal
    scores implementation-defined (observed: it differs between runs of different sizes).  Here the ascending sort is
    stable, i.e. of equal scores the LARGER flat index is taken first; the fixture holds no exact ties.
"""
import numpy as np

from .targets import CLASSES, gt_standup_2d

SCORE_THRES, NMS_THRES, NMS_POST_TOPK = 0.96, 0.1, 20          # config.py:95-98


def deltas_to_boxes_3d(deltas, anchors, cls_name="Car"):
    """utils.py:476-489: deltas (B,14,h,w) float32 (any shape with B leading), anchors (h,w,2,7) float64 -> (B,N,7) float32"""
    a = anchors.reshape(-1, 7)
    d = deltas.reshape(deltas.shape[0], -1, 7)
    diag = np.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2)
    out = np.zeros_like(d)
    out[..., [0, 1]] = d[..., [0, 1]] * diag[:, np.newaxis] + a[..., [0, 1]]
    out[..., [2]] = d[..., [2]] * CLASSES[cls_name]["h"] + a[..., [2]]
    out[..., [3, 4, 5]] = np.exp(d[..., [3, 4, 5]]) * a[..., [3, 4, 5]]
    out[..., 6] = d[..., 6] + a[..., 6]
    return out


def nms(boxes, scores, overlap=0.5, top_k=200):
    """utils.py:492-553 on NumPy arrays: boxes (M,4) float64 [x1,y1,x2,y2], scores (M,) float32 -> kept indices"""
    keep = []
    if boxes.size == 0:
        return np.zeros(0, dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    idx = np.argsort(scores, kind="stable")[-top_k:]
    while idx.size > 0:
        i = idx[-1]
        keep.append(i)
        if idx.size == 1:
            break
        idx = idx[:-1]
        xx1 = np.maximum(x1[idx], x1[i])
        yy1 = np.maximum(y1[idx], y1[i])
        xx2 = np.minimum(x2[idx], x2[i])
        yy2 = np.minimum(y2[idx], y2[i])
        w = np.maximum(xx2 - xx1, 0.0)
        h = np.maximum(yy2 - yy1, 0.0)
        inter = w * h
        union = (area[idx] - inter) + area[i]
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        idx = idx[iou <= overlap]
    return np.array(keep, dtype=np.int64)


def filter_boxes(probs, boxes_3d):
    """model.py:28-57: per sample score filter (>= SCORE_THRES), stand-up rectangles of the survivors, NMS ->
    ([boxes (n_i,7) float32], [scores (n_i,) float32])"""
    ret_b, ret_s = [], []
    for b in range(probs.shape[0]):
        idx = np.where(probs[b, :] >= SCORE_THRES)[0]
        tb, ts = boxes_3d[b, idx, ...], probs[b, idx]
        b2 = np.zeros((tb.shape[0], 7))
        b2[:, [0, 1, 4, 5, 6]] = tb[:, [0, 1, 4, 5, 6]]
        standup = gt_standup_2d(b2).astype(np.float64)
        keep = nms(standup, ts, NMS_THRES, NMS_POST_TOPK)
        ret_b.append(tb[keep, ...])
        ret_s.append(ts[keep])
    return ret_b, ret_s


def predict_boxes(probs, deltas, anchors, cls_name="Car"):
    """model.py:380-389: probs (B,2,h,w), deltas (B,14,h,w) float32 -> ([boxes], [scores]) after the NMS"""
    boxes = deltas_to_boxes_3d(deltas, anchors, cls_name)
    return filter_boxes(probs.reshape((probs.shape[0], -1)), boxes)
