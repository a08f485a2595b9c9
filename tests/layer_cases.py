"""Single-layer parity cases shared by the oracle (CPU) and HIP (GPU) tests.
Same table, seeds and closed-form weights as tools/gen_golden.py:LAYER_CASES."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as tr  # noqa: E402

LAYER_CASES = [
    # name, kind, dim, cin, cout, k, stride, pad, input spatial   (model.py:207-254 variants)
    ("c3_s211_p111", "conv", 3, 128, 64, 3, (2, 1, 1), (1, 1, 1), (10, 8, 12)),
    ("c3_s111_p011", "conv", 3, 64, 64, 3, (1, 1, 1), (0, 1, 1), (5, 8, 12)),
    ("c3_s211_p111_d3", "conv", 3, 64, 64, 3, (2, 1, 1), (1, 1, 1), (3, 8, 12)),
    ("c2_s1", "conv", 2, 128, 128, 3, (1, 1), (1, 1), (8, 12)),
    ("c2_s2", "conv", 2, 128, 128, 3, (2, 2), (1, 1), (8, 12)),
    ("c2_s2_256", "conv", 2, 128, 256, 3, (2, 2), (1, 1), (8, 12)),
    ("c2_s1_256", "conv", 2, 256, 256, 3, (1, 1), (1, 1), (6, 5)),
    ("d_k3s1", "deconv", 2, 128, 256, 3, (1, 1), (1, 1), (8, 12)),
    ("d_k2s2", "deconv", 2, 128, 256, 2, (2, 2), (0, 0), (4, 6)),
    ("d_k4s4", "deconv", 2, 256, 256, 4, (4, 4), (0, 0), (2, 3)),
    ("head_2", "head", 2, 768, 2, 1, (1, 1), (0, 0), (8, 12)),
    ("head_14", "head", 2, 768, 14, 1, (1, 1), (0, 0), (8, 12)),
]


def _seeded(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def build_layer_state(i, case):
    """state dict under prefix 'L' with the generator's closed-form values."""
    name, kind, dim, cin, cout, k, s, p, sp = case
    if kind == "deconv":
        wshape, wkey, fan = (cin, cout, k, k), "deconv", cin
    else:
        wshape, wkey, fan = (cout, cin) + (k,) * dim, "conv", cin * k ** dim
    sd = {f"L.{wkey}.weight": tr._fill(wshape, 200 + i, 1.0 / np.sqrt(fan)),
          f"L.{wkey}.bias": tr._fill((cout,), 300 + i, 0.1)}
    if kind != "head":
        sd.update({"L.batch_norm.weight": 1.0 + tr._fill((cout,), 400 + i, 0.2),
                   "L.batch_norm.bias": tr._fill((cout,), 500 + i, 0.1),
                   "L.batch_norm.running_mean": torch.zeros(cout),
                   "L.batch_norm.running_var": torch.ones(cout),
                   "L.batch_norm.num_batches_tracked": torch.zeros((), dtype=torch.long)})
    return sd


def layer_input(i, case):
    return _seeded((2, case[3]) + case[8], 600 + i)


def layer_upstream(i, case, shape):
    return _seeded(shape, 700 + i)
