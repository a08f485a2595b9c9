"""ORACLE (test infrastructure, not the product path): NumPy restatement of the reference's RPN target generation —
`generate_anchors` (utils.py:104-130), `label_to_gt_box_3d` / `camera_to_lidar_box` / `angle_in_limit`
(utils.py:133-212), `center_to_corner_box_2d` (utils.py:240-252, 283-330), `corner_to_standup_box2d` (utils.py:230-238),
`anchor_to_standup_box2d` (utils.py:213-227), `bbox_iou` (utils.py:344-373) and `generate_targets` (utils.py:376-473).

Pinned by tests/golden/targets_*.npz, which tools/gen_golden.py wrote by calling the imported reference
(tests/test_oracle_targets.py).  Reference quirks kept on purpose:
  * anchor stand-up boxes have x2 == x1 and y2 == y1 (utils.py:219-220, 224-225 subtract where they should add);
  * the union term of the IoU uses (y1 - x1 + 1) * (y2 - y1 + 1) of the ANCHOR box (utils.py:367);
  * `+ 1` pixel conventions on metric boxes; an anchor may be positive (best anchor of a box) and negative at once.
Arithmetic types follow what the reference's expressions evaluate to under NumPy >= 2 (NEP 50), the NumPy of this
image and of the fixtures: the IoU terms are float32 (float32 scalars with Python ints), the union is widened to a
Python float and the quotient rounded back to float32; the regression targets are float64.
"""
import numpy as np

# config.py:36-92 (Car / Pedestrian / Cyclist), config.py:99-111 (mean KITTI calibration)
CLASSES = {
    "Car": dict(x=(0.0, 70.4), y=(-40.0, 40.0), fw=176, fh=200, l=3.9, w=1.6, h=1.56, z=-1.0 - 1.56 / 2, pos=0.6, neg=0.45),
    "Pedestrian": dict(x=(0.0, 48.0), y=(-20.0, 20.0), fw=120, fh=100, l=0.8, w=0.6, h=1.73, z=-0.6 - 1.73 / 2, pos=0.5,
                       neg=0.35),
    "Cyclist": dict(x=(0.0, 48.0), y=(-20.0, 20.0), fw=120, fh=100, l=1.76, w=0.6, h=1.73, z=-0.6 - 1.73 / 2, pos=0.5,
                    neg=0.35),
}
T_VELO_2_CAM = np.array([[7.49916597e-03, -9.99971248e-01, -8.65110297e-04, -6.71807577e-03],
                         [1.18652889e-02, 9.54520517e-04, -9.99910318e-01, -7.33152811e-02],
                         [9.99882833e-01, 7.49141178e-03, 1.18719929e-02, -2.78557062e-01],
                         [0, 0, 0, 1]])
R_RECT_0 = np.array([[0.99992475, 0.00975976, -0.00734152, 0],
                     [-0.0097913, 0.99994262, -0.00430371, 0],
                     [0.00729911, 0.0043753, 0.99996319, 0],
                     [0, 0, 0, 1]])


def generate_anchors(cls_name="Car"):
    """utils.py:104-130 -> (fh, fw, 2, 7) float64 [x, y, z, h, w, l, r]; r = 0 and pi/2."""
    c = CLASSES[cls_name]
    x = np.linspace(c["x"][0], c["x"][1], c["fw"])
    y = np.linspace(c["y"][0], c["y"][1], c["fh"])
    cx, cy = np.meshgrid(x, y)
    cx = np.tile(cx[..., np.newaxis], 2)
    cy = np.tile(cy[..., np.newaxis], 2)
    one = np.ones_like(cx)
    r = np.ones_like(cx)
    r[..., 0] = 0
    r[..., 1] = 90 / 180 * np.pi
    return np.stack([cx, cy, one * c["z"], one * c["h"], one * c["w"], one * c["l"], r], axis=-1)


def angle_in_limit(angle):
    """utils.py:133-144"""
    while angle >= np.pi / 2:
        angle -= np.pi
    while angle < -np.pi / 2:
        angle += np.pi
    if abs(angle + np.pi / 2) < 5 / 180 * np.pi:
        angle = np.pi / 2
    return angle


def camera_to_lidar_box(boxes):
    """utils.py:147-174: (x,y,z,h,w,l,ry) camera -> (x,y,z,h,w,l,rz) lidar with the mean calibration"""
    out = []
    rinv, tinv = np.linalg.inv(R_RECT_0), np.linalg.inv(T_VELO_2_CAM)
    for x, y, z, h, w, l, ry in boxes:
        p = np.matmul(tinv, np.matmul(rinv, np.array([x, y, z, 1])))[:3]
        out.append([p[0], p[1], p[2], h, w, l, angle_in_limit(-ry - np.pi / 2)])
    return np.array(out).reshape(-1, 7)


def label_to_gt_box_3d(labels, cls_name="Car", coordinate="lidar"):
    """utils.py:178-210: KITTI label lines per sample -> list of (G_i, 7) float64 boxes"""
    acc = {"Car": ["Car", "Van"], "Pedestrian": ["Pedestrian"], "Cyclist": ["Cyclist"]}.get(cls_name, [])
    boxes = []
    for label in labels:
        rows = []
        for line in label:
            a = line.split()
            if a[0] in acc or acc == []:
                h, w, l, x, y, z, r = [float(v) for v in a[-7:]]
                rows.append(np.array([x, y, z, h, w, l, r]))
        rows = np.array(rows)
        if coordinate == "lidar":
            rows = camera_to_lidar_box(rows)
        boxes.append(np.array(rows).reshape(-1, 7))
    return boxes


def gt_standup_2d(gt):
    """center_to_corner_box_2d (utils.py:240-252 via 283-330: float64 rotation, float32 corner storage) followed by
    corner_to_standup_box2d (utils.py:230-238), on lidar boxes (G,7) -> (G,4) float32 [x1,y1,x2,y2]"""
    G = gt.shape[0]
    out = np.zeros((G, 4))
    for i in range(G):
        x, y, w, l, yaw = gt[i, 0], gt[i, 1], gt[i, 4], gt[i, 5], gt[i, 6]
        tracklet = np.array([[-l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2],
                             [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2],
                             [0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0]])      # h = 0 here (utils.py:246-247 zero the size's h)
        rot = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
        corners = (np.dot(rot, tracklet) + np.tile(np.array([x, y, 0.0]), (8, 1)).T).transpose().astype(np.float32)
        c = corners[0:4, 0:2]
        out[i] = [c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()]
    return out.astype(np.float32)


def anchor_standup_2d(anchors):
    """utils.py:213-227 on anchors[:, [0,1,4,5]] = (x, y, w, l): x2/y2 come out EQUAL to x1/y1 (the quirk)."""
    a = anchors.reshape(-1, 7)[:, [0, 1, 4, 5]]
    s = np.zeros_like(a)
    s[::2, 0] = a[::2, 0] - a[::2, 3] / 2
    s[::2, 1] = a[::2, 1] - a[::2, 2] / 2
    s[::2, 2] = a[::2, 0] - a[::2, 3] / 2
    s[::2, 3] = a[::2, 1] - a[::2, 2] / 2
    s[1::2, 0] = a[1::2, 0] - a[1::2, 2] / 2
    s[1::2, 1] = a[1::2, 1] - a[1::2, 3] / 2
    s[1::2, 2] = a[1::2, 0] - a[1::2, 2] / 2
    s[1::2, 3] = a[1::2, 1] - a[1::2, 3] / 2
    return s.astype(np.float32)


def bbox_iou(box1, box2):
    """utils.py:344-373, vectorised with the scalar loop's arithmetic types: float32 terms, the union widened to
    float64 by `float(...)`, quotient stored as float32.  box1 (N,4), box2 (K,4) float32 -> (N,K) float32."""
    one = np.float32(1)
    b1, b2 = box1[:, None, :], box2[None, :, :]
    area2 = (b2[..., 2] - b2[..., 0] + one) * (b2[..., 3] - b2[..., 1] + one)
    iw = np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0]) + one
    ih = np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1]) + one
    ua = ((b1[..., 1] - b1[..., 0] + one) * (b1[..., 3] - b1[..., 1] + one) + area2 - iw * ih).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = ((iw * ih).astype(np.float64) / ua).astype(np.float32)
    return np.where((iw > 0) & (ih > 0), q, np.float32(0)).astype(np.float32)


def generate_targets_from_boxes(gt_boxes, feature_map_shape, anchors, cls_name="Car"):
    """utils.py:385-473 given the per-sample lidar boxes -> (pos (B,h,w,2), neg (B,h,w,2), targets (B,h,w,14)) float64"""
    c = CLASSES[cls_name]
    B = len(gt_boxes)
    a = anchors.reshape(-1, 7)
    diag = np.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2)
    pos = np.zeros((B, *feature_map_shape, 2))
    neg = np.zeros((B, *feature_map_shape, 2))
    tgt = np.zeros((B, *feature_map_shape, 14))
    a2d = anchor_standup_2d(anchors)
    for b in range(B):
        gt = gt_boxes[b]
        iou = bbox_iou(a2d, gt_standup_2d(gt))
        id_max = np.argmax(iou.T, axis=1)
        id_max_gt = np.arange(iou.T.shape[0])
        keep = iou.T[id_max_gt, id_max] > 0
        id_max, id_max_gt = id_max[keep], id_max_gt[keep]
        id_pos, id_pos_gt = np.where(iou > c["pos"])
        id_neg = np.where(np.sum(iou < c["neg"], axis=1) == iou.shape[1])[0]
        id_pos = np.concatenate([id_pos, id_max])
        id_pos_gt = np.concatenate([id_pos_gt, id_max_gt])
        id_pos, first = np.unique(id_pos, return_index=True)
        id_pos_gt = id_pos_gt[first]
        iy, ix, iz = np.unravel_index(id_pos, (*feature_map_shape, 2))
        pos[b, iy, ix, iz] = 1
        g, an = gt[id_pos_gt], a[id_pos]
        tgt[b, iy, ix, iz * 7 + 0] = (g[:, 0] - an[:, 0]) / diag[id_pos]
        tgt[b, iy, ix, iz * 7 + 1] = (g[:, 1] - an[:, 1]) / diag[id_pos]
        tgt[b, iy, ix, iz * 7 + 2] = (g[:, 2] - an[:, 2]) / c["h"]
        tgt[b, iy, ix, iz * 7 + 3] = np.log(g[:, 3] / an[:, 3])
        tgt[b, iy, ix, iz * 7 + 4] = np.log(g[:, 4] / an[:, 4])
        tgt[b, iy, ix, iz * 7 + 5] = np.log(g[:, 5] / an[:, 5])
        tgt[b, iy, ix, iz * 7 + 6] = g[:, 6] - an[:, 6]
        iy, ix, iz = np.unravel_index(id_neg, (*feature_map_shape, 2))
        neg[b, iy, ix, iz] = 1
    return pos, neg, tgt


def generate_targets(labels, feature_map_shape, anchors, cls_name="Car", coordinate="lidar"):
    """utils.py:376-473"""
    return generate_targets_from_boxes(label_to_gt_box_3d(labels, cls_name, coordinate), feature_map_shape, anchors, cls_name)
