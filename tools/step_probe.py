"""When does what finish inside one train step?  Wraps voxelnet_amd._lib.call so that a timing event is recorded, on the
stream the call was issued on, behind every library call of the step (VFE forward, the two vn_net_prepare calls on the
side stream, vn_net_forward, loss, vn_net_backward, VFE backward, optimizer), runs 30 car-config train steps on static
pre-voxelized input and prints, for the last 10, the median offset of every probe from the step's first one.  Unlike the
executor's timing mode this leaves the launches inside the native calls untouched (one event per CALL, not per launch).

    python tools/step_probe.py > gpurun_out/step_probe.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from voxelnet_amd import _lib, model as M, synth  # noqa: E402
from voxelnet_amd.config import grid_config  # noqa: E402
from voxelnet_amd.optim import ClipSGD  # noqa: E402
from voxelnet_amd.voxelize import voxelize_device  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
grid = grid_config("Car")
frames = synth.workload_frames(2)
feats, coords = [], []
for b, f in enumerate(frames):
    fb, cb, _ = voxelize_device(torch.from_numpy(f).to(dev), grid, b, coord_cols=4)
    feats.append(fb)
    coords.append(cb)
labels = [synth.synth_labels("Car", 6, seed=70 + b) for b in range(2)]
M.set_precision("bf16")
m = M.RPN3D("Car").to(dev).train()
opt = ClipSGD(m.parameters(), lr=0.01, max_norm=5.0)

probes = []          # (step, name, event)
state = {"step": -1}
orig_call = _lib.call
STREAM_ARG = {"vn_net_forward": -2, "vn_net_backward": -2}


def stream_of(name, args):
    a = args[STREAM_ARG.get(name, -1)]
    ptr = a.value if isinstance(a, ctypes.c_void_p) else (a.cuda_stream if hasattr(a, "cuda_stream") else a)
    return ptr


def probed_call(name, *args):
    orig_call(name, *args)
    if state["step"] < 0 or not name.startswith(("vn_net_", "vn_vfe_", "vn_rpn_loss", "vn_clip", "vn_opt", "vn_rpn_targets", "vn_cast_rows")):
        return
    try:
        ptr = stream_of(name, args)
        st = torch.cuda.ExternalStream(int(ptr)) if ptr else torch.cuda.default_stream()
    except Exception:   # noqa: BLE001 - a call without a stream argument
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(st)
    n = sum(1 for s, nm, _ in probes if s == state["step"] and nm.split("#")[0] == name)
    probes.append((state["step"], f"{name}#{n}" if n else name, ev))


_lib.call = probed_call
for mod in list(sys.modules.values()):          # modules that did `from . import _lib` call _lib.call through the module: patched above
    pass
STEPS = 30
starts = []
for i in range(STEPS):
    state["step"] = i
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    starts.append(e0)
    out = m((None, labels, feats, None, coords, None, None), dev)
    out[2].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
names = []
for s, nm, _ in probes:
    if s == STEPS - 1 and nm not in names:
        names.append(nm)
print("median over the last 10 steps: offset (us) of the END of each library call from the first launch of its step")
rows = []
for nm in names:
    offs = [starts[s].elapsed_time(ev) * 1e3 for s, n2, ev in probes if n2 == nm and s >= STEPS - 10]
    if offs:
        rows.append((float(np.median(offs)), nm))
step_len = np.median([starts[i].elapsed_time(starts[i + 1]) * 1e3 for i in range(STEPS - 10, STEPS - 1)])
for off, nm in sorted(rows):
    print(f"  {off:9.1f}  {nm}")
print(f"  {step_len:9.1f}  (next step's first launch)")
