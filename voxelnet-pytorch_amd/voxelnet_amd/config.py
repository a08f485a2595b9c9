"""Grid / network constants of the hot path, restated from the reference's
config tree (/root/reference/voxelnet/config.py:36-98) and the literals in
utils.py:24-33.  Plain dataclass instead of yacs: kernels take every value as a
runtime argument."""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class GridConfig:
    name: str
    D: int
    H: int
    W: int
    vz: float
    vy: float
    vx: float
    ox: float      # lidar_coord added to x (utils.py:27,32)
    oy: float
    oz: float
    T: int         # POINTS_PER_VOXEL (config.py:47,71)
    block1_stride: int   # model.py:212-227: 2 for Car, 1 otherwise

    @property
    def dims(self):
        return (self.D, self.H, self.W)

    @property
    def cells(self):
        return self.D * self.H * self.W


CAR = GridConfig("Car", 10, 400, 352, 0.4, 0.2, 0.2, 0.0, 40.0, 3.0, 35, 2)
PED = GridConfig("Pedestrian", 10, 200, 240, 0.4, 0.2, 0.2, 0.0, 20.0, 3.0, 45, 1)
CYC = replace(PED, name="Cyclist")

# config.py:15-23 (TRAIN) and 95-98 (RPN)
ALPHA, BETA, SIGMA = 1.5, 1.0, 3.0
LR, GRADIENT_CLIP = 0.01, 5.0


def grid_config(target="Car", T=None, **override):
    g = {"Car": CAR, "Pedestrian": PED, "Cyclist": CYC}.get(target)
    if g is None:
        raise ValueError(f"unknown target class {target!r}")
    if T is not None:
        g = replace(g, T=int(T))
    return replace(g, **override) if override else g
