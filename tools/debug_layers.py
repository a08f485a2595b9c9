"""Debug: per-layer activations of the HIP middle net vs the oracle (runs on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from dataclasses import replace
from oracle import torch_ref as tr
from voxelnet_amd import model as M, engine as E, net as N

g = np.load(os.path.join(ROOT, "tests/golden/middle_tiny_car.npz"))
feats = torch.from_numpy(g["features"]); coords = torch.from_numpy(g["coords"])
lens = [int(x) for x in g["feat_lens"]]
fl, cl = list(torch.split(feats, lens)), list(torch.split(coords, lens))
sd = tr.make_state_dict("Car")
dense = tr.feature_net(fl, cl, sd, (10, 16, 24), True)
taps = {}
sd2 = tr.make_state_dict("Car")
# use stats-updated-free copy for the oracle taps
pr, rr = tr.middle_rpn(dense, sd2, "Car", True, taps)

M.set_precision("exact")
m = M.RPN3D("Car"); m.load_state_dict(tr.make_state_dict("Car")); m.feature_net._grid = replace(m.feature_net._grid, H=16, W=24)
m = m.to("cuda:0").train()
mid = m.middle_rpn
names, P, Bf, flat = M._collect_middle(mid)
P = M._detached(P); P["heads"] = M._heads_params([f.detach() for f in flat])
dr = E.new_rows(2, (10, 16, 24), 128, torch.bfloat16, True, "cuda:0")
xc = dense.cuda().contiguous()
M._lib.call("vn_cast_rows", xc.data_ptr(), 0, 128, 2 * 10 * 16 * 24, 128, dr.ptr(), 1, 256, 128, E.stream())
prob, reg, st = N.middle_forward(dr, P, Bf, 2, True, True)

def act(a, dim):
    if a.lo_off and a.lo_off != a.C:      # slice of the concat buffer
        hi = E.rows_to_nchw(E.Rows(a.t, a.C), dim)
        full = a.t._base if a.t._base is not None else a.t
        off = a.t.storage_offset() - full.storage_offset()
        lo_t = torch.as_strided(full, a.t.shape, a.t.stride(), a.t.storage_offset() + a.lo_off)
        return (hi + E.rows_to_nchw(E.Rows(lo_t, a.C), dim)).cpu()
    return M._act_to_nchw(a, dim).cpu()

for name in names:
    s = st.layers[name]
    ref = taps[name].detach()
    a = s.a
    if name == "middle_layer.2":
        got = act(a, 2).reshape(2, 2, 64, 16, 24).permute(0, 2, 1, 3, 4)  # stored d*64+c
    elif name.startswith("middle"):
        got = act(a, 3)
    else:
        got = act(a, 2)
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    print(f"{name:16s} shape {tuple(ref.shape)} err {err:.3e}")
print("prob", (prob.cpu() - pr).abs().max().item(), "reg", (reg.cpu() - rr).abs().max().item() / rr.abs().max().item())

# ---- isolated: each layer fed with the ORACLE's input activation
print("isolated per-layer errors (oracle input -> HIP layer -> vs oracle output)")
specs = dict(N.layer_table(2))
prev = {"middle_layer.0": dense.permute(0, 4, 1, 2, 3).contiguous()}
order = names
inputs = {}
x = dense.permute(0, 4, 1, 2, 3).contiguous()
for i, name in enumerate(order):
    if name == "middle_layer.0": inp = x
    elif name == "middle_layer.1": inp = taps["middle_layer.0"]
    elif name == "middle_layer.2": inp = taps["middle_layer.1"]
    elif name == "block1.0": inp = None
    elif name == "deconv1": inp = taps["block1.4"]
    elif name == "block2.0": inp = taps["block1.4"]
    elif name == "deconv2": inp = taps["block2.5"]
    elif name == "block3.0": inp = taps["block2.5"]
    elif name == "deconv3": inp = taps["block3.5"]
    else:
        blk, idx = name.split("."); inp = taps[f"{blk}.{int(idx)-1}"]
    if inp is None: continue
    sp = specs[name]
    sdx = tr.make_state_dict("Car")
    mm = M.RPN3D("Car"); mm.load_state_dict(sdx); mm = mm.to("cuda:0").train()
    n2, P2, B2, f2 = M._collect_middle(mm.middle_rpn); P2 = M._detached(P2)
    xr = E.nchw_to_rows(inp.detach().cuda(), True)
    a, s_ = E.layer_forward(sp, xr, P2[name], B2[name], True, True)
    got = M._act_to_nchw(a, sp.dim).cpu()
    ref = taps[name].detach()
    print(f"{name:16s} err {(got-ref).abs().max().item()/ref.abs().max().item():.3e}")
