"""Time the device RPN target generation (vn_rpn_targets + its host half) on the fixture's label sets.
usage: python tools/bench_targets.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np
import torch
from voxelnet_amd import targets as T

g = np.load(os.path.join(ROOT, "tests", "golden", "targets_car.npz"))
labels = [[str(s) for s in g["labels0"]], [str(s) for s in g["labels3"]]]      # 7 + 14 boxes, B = 2
gen = T.TargetGenerator("Car", "cuda:0")
boxes = T.label_to_gt_box_3d(labels, "Car")
for _ in range(3):
    gen(labels)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    T.label_to_gt_box_3d(labels, "Car")
print(f"host: label parsing + camera->lidar        {(time.perf_counter() - t) / 50 * 1e3:.3f} ms")
t = time.perf_counter()
for _ in range(50):
    out = gen.from_boxes(boxes)
torch.cuda.synchronize()
print(f"host stand-up boxes + H2D + 3 launches (wall) {(time.perf_counter() - t) / 50 * 1e3:.3f} ms")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    out = gen.from_boxes(boxes)
e.record()
torch.cuda.synchronize()
print(f"device stream time per call               {s.elapsed_time(e) / 50:.3f} ms")
