import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "voxelnet-pytorch_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from layer_cases import LAYER_CASES, build_layer_state, layer_input
from voxelnet_amd import engine as E
from oracle import torch_ref as tr
import torch.nn.functional as F
case = LAYER_CASES[0]; idx = 0
name, kind, dim, cin, cout, k, s, p, sp = case
sd = build_layer_state(idx, case)
dev = "cuda:0"
x = layer_input(idx, case)
ref = F.conv3d(x.double(), sd["L.conv.weight"].double(), sd["L.conv.bias"].double(), s, p)
spec = E.LayerSpec(name, 3, cin, cout, (k, k, k), tuple(s), tuple(p))
for mode in ["fp32", "bf16x3", "bf16"]:
    P = {"weight": sd["L.conv.weight"].to(dev), "bias": sd["L.conv.bias"].to(dev), "gamma": sd["L.batch_norm.weight"].to(dev), "beta": sd["L.batch_norm.bias"].to(dev)}
    Bf = {"running_mean": torch.zeros(cout, device=dev), "running_var": torch.ones(cout, device=dev)}
    a, st = E.layer_forward(spec, E.nchw_to_rows(x.to(dev), mode), P, Bf, True, mode)
    y = E.rows_to_nchw(st.y, 3).cpu().double()
    d = (y - ref).abs()
    print(mode, "conv y err max", (d.max() / ref.abs().max()).item(), "mean", (d.mean() / ref.abs().mean()).item())
    bad = (d > 1e-3 * ref.abs().max()).nonzero()
    print("  bad count", bad.shape[0], bad[:10].tolist())
print("---- BN output")
import torch.nn.functional as F
g = np.load(os.path.join(ROOT, "tests/golden/layers_tiny.npz"))
for mode in ["fp32", "bf16x3"]:
    P = {"weight": sd["L.conv.weight"].to(dev), "bias": sd["L.conv.bias"].to(dev), "gamma": sd["L.batch_norm.weight"].to(dev), "beta": sd["L.batch_norm.bias"].to(dev)}
    Bf = {"running_mean": torch.zeros(cout, device=dev), "running_var": torch.ones(cout, device=dev)}
    a, st = E.layer_forward(spec, E.nchw_to_rows(x.to(dev), mode), P, Bf, True, mode)
    C = a.C
    hi = E.rows_to_nchw(E.Rows(a.t[..., :C], C), 3).cpu()
    lo = E.rows_to_nchw(E.Rows(a.t[..., a.lo_off:a.lo_off + C], C), 3).cpu() if a.lo_off else 0
    got = hi + lo
    ref_y = torch.from_numpy(g[name + ".y"])
    d = (got - ref_y).abs()
    print(mode, "act err", (d.max() / ref_y.abs().max()).item(), "hi-only err", ((hi - ref_y).abs().max() / ref_y.abs().max()).item())
    stats = st.stats.cpu()
    y = E.rows_to_nchw(st.y, 3).cpu()
    mean_ref = y.transpose(0, 1).reshape(64, -1).double().mean(1)
    var_ref = y.transpose(0, 1).reshape(64, -1).double().var(1, unbiased=False)
    print("  mean err", (stats[:64].double() - mean_ref).abs().max().item(), "invstd err", (stats[64:128].double() - 1 / torch.sqrt(var_ref + 1e-5)).abs().max().item())
    bad = (d > 1e-3 * ref_y.abs().max()).nonzero()
    print("  bad", bad.shape[0], bad[:8].tolist())
