"""Seeded synthetic KITTI-shaped point clouds (SURVEY.md §8d "voxel-first" generator).

The generator controls K and overflow directly: it picks K0 distinct voxels,
draws a point count per voxel, places points strictly inside each voxel, appends
out-of-range points (including values in (-0.2, 0) that separate floor from
truncation), and permutes.  numpy only — used by tests, bench.py and
tools/gen_golden.py; it never touches the GPU.
"""
import numpy as np

from .config import grid_config

_Z_WEIGHTS = np.array([1, 3, 6, 6, 4, 2, 1, 1, .5, .5])


def frame_seed(config_id, frame):
    return 20240000 + 1000 * int(config_id) + int(frame)


def synth_cloud(target="Car", k0=6000, seed=0, mean_extra=2.3, T=None, grid=None,
                overflow_frac=0.005, outside_frac=0.05):
    """Returns (N,4) float32 [x,y,z,reflectance] in final (already permuted) order."""
    g = grid if grid is not None else grid_config(target, T=T)
    D, H, W, Tn = g.D, g.H, g.W, g.T
    rng = np.random.default_rng(seed)
    k0 = int(min(k0, D * H * W))
    # 1. distinct voxels: denser near the sensor (small x index), z-slab weights
    zw = np.resize(_Z_WEIGHTS, D).astype(np.float64)
    xw = 1.0 / (1.0 + np.arange(W) / 16.0)
    p_zx = np.outer(zw / zw.sum(), xw / xw.sum())                # (D,W)
    p = np.broadcast_to(p_zx[:, None, :] / H, (D, H, W)).reshape(-1)
    lin = rng.choice(D * H * W, size=k0, replace=False, p=p / p.sum())
    z, y, x = lin // (H * W), (lin // W) % H, lin % W
    # 2. per-voxel point counts
    n = 1 + np.floor(rng.exponential(mean_extra, size=k0)).astype(np.int64)
    over = rng.random(k0) < overflow_frac
    n[over] = Tn + 1 + rng.integers(0, Tn + 1, size=int(over.sum()))
    rep = np.repeat(np.arange(k0), n)
    # 3. points strictly inside their voxel (1e-3 margin of the voxel size)
    u = rng.uniform(1e-3, 1 - 1e-3, size=(rep.size, 3))
    px = (x[rep] + u[:, 0]) * g.vx - g.ox
    py = (y[rep] + u[:, 1]) * g.vy - g.oy
    pz = (z[rep] + u[:, 2]) * g.vz - g.oz
    refl = np.round(rng.uniform(0, 1, size=rep.size), 2)
    pts = np.stack([px, py, pz, refl], 1)
    # 4. out-of-range points
    m = int(outside_frac * rep.size)
    if m > 0:
        out = np.stack([rng.uniform(0, W * g.vx, m), rng.uniform(-g.oy, H * g.vy - g.oy, m),
                        rng.uniform(-g.oz, D * g.vz - g.oz, m), np.round(rng.uniform(0, 1, m), 2)], 1)
        kind = rng.integers(0, 5, m)
        out[kind == 0, 0] = -rng.uniform(0, 0.2, int((kind == 0).sum()))          # x in (-0.2,0)
        out[kind == 1, 1] = (H * g.vy - g.oy) + rng.uniform(0, 5, int((kind == 1).sum()))
        out[kind == 2, 1] = -g.oy - rng.uniform(0, 0.2, int((kind == 2).sum()))   # y just below range
        out[kind == 3, 2] = -g.oz - rng.uniform(0, 0.4, int((kind == 3).sum()))   # z just below range
        out[kind == 4, 2] = (D * g.vz - g.oz) + rng.uniform(0, 2, int((kind == 4).sum()))
        pts = np.concatenate([pts, out], 0)
    # 5. the permutation *is* the shuffle
    pts = pts[rng.permutation(pts.shape[0])]
    return np.ascontiguousarray(pts, dtype=np.float32)


WORKLOADS = {
    # BASELINE.json configs -> generator arguments
    1: dict(target="Car", k0=6000, mean_extra=2.3, T=35, batch=1),
    2: dict(target="Car", k0=6000, mean_extra=2.3, T=35, batch=2),
    3: dict(target="Pedestrian", k0=5000, mean_extra=2.3, T=45, batch=2),
    4: dict(target="Car", k0=6000, mean_extra=2.3, T=35, batch=2),   # per GPU
    5: dict(target="Car", k0=40000, mean_extra=6.5, T=64, batch=4),
}


def workload_frames(config_id, batch=None, frame0=0):
    w = WORKLOADS[config_id]
    b = w["batch"] if batch is None else batch
    return [synth_cloud(w["target"], w["k0"], frame_seed(config_id, frame0 + f), w["mean_extra"], w["T"])
            for f in range(b)]


def synth_labels(target="Car", n_objects=6, seed=0):
    """KITTI label lines of one synthetic frame: n_objects boxes of the class inside the crop plus one DontCare line —
    what the dataset hands to RPN3D.forward as x[1][i] (the targets are generated from them on the device, model.py:309)"""
    from .targets import CLASS_CFG, lidar_box_to_label_line
    c = CLASS_CFG[target]
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(int(n_objects)):
        box = [rng.uniform(c["x"][0] + 4, c["x"][1] - 4), rng.uniform(c["y"][0] + 4, c["y"][1] - 4), c["z"] + rng.uniform(-0.2, 0.2),
               c["h"] * rng.uniform(0.9, 1.1), c["w"] * rng.uniform(0.9, 1.1), c["l"] * rng.uniform(0.9, 1.1), rng.uniform(-1.5, 1.5)]
        lines.append(lidar_box_to_label_line(target, box))
    lines.append("DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10")
    return lines
