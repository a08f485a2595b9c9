"""Functional forward/backward of the middle layers + RPN (MiddleConvNet,
/root/reference/voxelnet/model.py:202-281) on channels-last rows, built from
engine.layer_forward / layer_backward.  No autograd here: model.py wraps these in
torch.autograd.Function so the reference's train loop (train.py:148-155) runs unchanged.
"""
import torch

from . import _lib
from . import engine as E
from .engine import Rows, spec2, spec3

HEADS = spec2("heads", 768, 16, 1, (1, 1), (0, 0), bn=False, relu=False)


_TABLES = {}


def layer_table(block1_stride):
    """model.py:206-254 in execution order: [(name, spec)], names = state_dict prefixes (cached: read-only)."""
    hit = _TABLES.get(block1_stride)
    if hit is None:
        hit = _TABLES[block1_stride] = _layer_table(block1_stride)
    return hit


def _layer_table(block1_stride):
    s1 = (block1_stride, block1_stride)
    t = [("middle_layer.0", spec3("middle_layer.0", 128, 64, 3, (2, 1, 1), (1, 1, 1))),
         ("middle_layer.1", spec3("middle_layer.1", 64, 64, 3, (1, 1, 1), (0, 1, 1))),
         ("middle_layer.2", spec3("middle_layer.2", 64, 64, 3, (2, 1, 1), (1, 1, 1)))]
    t += [("block1.0", spec2("block1.0", 128, 128, 3, s1, (1, 1), cin_fold=2))]
    t += [(f"block1.{i}", spec2(f"block1.{i}", 128, 128, 3, (1, 1), (1, 1))) for i in range(1, 5)]
    t += [("deconv1", spec2("deconv1", 128, 256, 3, (1, 1), (1, 1), transposed=True))]
    t += [("block2.0", spec2("block2.0", 128, 128, 3, (2, 2), (1, 1)))]
    t += [(f"block2.{i}", spec2(f"block2.{i}", 128, 128, 3, (1, 1), (1, 1))) for i in range(1, 6)]
    t += [("deconv2", spec2("deconv2", 128, 256, 2, (2, 2), (0, 0), transposed=True))]
    t += [("block3.0", spec2("block3.0", 128, 256, 3, (2, 2), (1, 1)))]
    t += [(f"block3.{i}", spec2(f"block3.{i}", 256, 256, 3, (1, 1), (1, 1))) for i in range(1, 6)]
    t += [("deconv3", spec2("deconv3", 256, 256, 4, (4, 4), (0, 0), transposed=True))]
    return t


class MiddleState:
    pass


def middle_forward(dense, P, Bf, block1_stride, training, mode, sparse=None):
    """dense: Rows (B,D,H,W,128[hi|lo]).  P[name] = {weight,bias,gamma,beta}; Bf[name] =
    {running_mean,running_var}; P['heads'] = {weight (16,768,1,1), bias (16)}.
    sparse = (coord (K,4) int64, vw_rows (K,128) operand dtype): the occupied sites of `dense`; the first
    Conv3d then runs in its sparse form (engine.first_layer_forward_sparse) and middle_backward returns the
    (K,128) voxel gradient instead of a dense grid gradient.
    Returns prob (B,2,h,w) after sigmoid, reg (B,14,h,w) fp32 NCHW, and the saved state."""
    specs = dict(layer_table(block1_stride))
    st = MiddleState()
    st.layers = {}
    st.block1_stride = block1_stride
    st.mode = mode
    st.x3 = bool(E.X3["on"])       # (the backward runs its products as this forward did: engine._x3_as_saved)
    split = E.is_split(mode)
    dev = dense.t.device
    B = dense.B

    def run(name, x, **kw):
        a, s = E.layer_forward(specs[name], x, P[name], Bf[name], training, mode, **kw)
        st.layers[name] = s
        return a

    st.sparse = sparse if (sparse is not None and not E.is_split(mode)) else None
    if st.sparse is not None:
        x, s0 = E.first_layer_forward_sparse(specs["middle_layer.0"], dense, st.sparse[0], P["middle_layer.0"],
                                             Bf["middle_layer.0"], training, mode)
        st.layers["middle_layer.0"] = s0
    else:
        x = run("middle_layer.0", dense)
    x = run("middle_layer.1", x)
    x = run("middle_layer.2", x, bev_out=True)           # -> (B,1,H,W,128) BEV rows
    for i in range(5):
        x = run(f"block1.{i}", x)
    x1 = x
    hf, wf = specs["deconv1"].out_dims(x1.dims)[1:]
    width = 768 * (2 if split else 1)
    cat = Rows(torch.empty((B, 1, hf, wf, width), dtype=E.act_dtype_of(mode), device=dev), 768, 768 if split else 0)

    def cat_slice(off):
        return Rows(cat.t[..., off:off + 256], 256, cat.lo_off)

    run("deconv1", x1, out=cat_slice(512))               # model.py:271-273: cat([d3, d2, d1])
    for i in range(6):
        x = run(f"block2.{i}", x)
    x2 = x
    run("deconv2", x2, out=cat_slice(256))
    for i in range(6):
        x = run(f"block3.{i}", x)
    run("deconv3", x, out=cat_slice(0))
    y, s = E.layer_forward(HEADS, cat, P["heads"], None, training, mode, y_dtype=torch.float32)
    st.layers["heads"] = s
    prob = E.rows_to_nchw(Rows(y.t[..., 0:2], 2), 2, sigmoid_first_n=2)
    reg = E.rows_to_nchw(Rows(y.t[..., 2:16], 14), 2)
    st.prob = prob.detach()     # an alias, NOT the returned tensor: saving an autograd.Function's own output on its ctx is a
                                # reference cycle that only the cyclic GC frees (720 MB of dense grid per step piled up)
    st.fmap = (hf, wf)
    return prob, reg, st


def middle_backward(st, d_prob, d_reg, P, need_dx=True, on_grads=None):
    """-> ({name: {weight,bias,gamma,beta}}, d_dense Rows (plain f32/bf16) or None).
    on_grads(layer_name, grads): called as soon as a layer's parameter gradients exist (DDP bucketing)."""
    with E._x3_as_saved(st):
        return _middle_backward(st, d_prob, d_reg, P, need_dx, on_grads)


def _middle_backward(st, d_prob, d_reg, P, need_dx=True, on_grads=None):
    mode = st.mode
    split = E.is_split(mode)
    L = st.layers
    dev = d_prob.device
    B = d_prob.shape[0]
    hf, wf = st.fmap
    G = {}
    w16 = 32 if split else 16
    d_rows = Rows(torch.empty((B, 1, hf, wf, w16), dtype=E.act_dtype_of(mode), device=dev), 16, 16 if split else 0)
    _lib.call("vn_heads_bwd", d_prob.contiguous().data_ptr(), d_reg.contiguous().data_ptr(), st.prob.data_ptr(), B,
              hf * wf, d_rows.ptr(), E._dt(d_rows.t), w16, int(split), E.stream())
    G["heads"], d_cat = E.layer_backward(L["heads"], d_rows, P["heads"], mode)
    if on_grads is not None:
        on_grads("heads", G["heads"])

    def dslice(off):
        return Rows(d_cat.t[..., off:off + 256], 256)

    def back(name, da, **kw):
        g, dx = E.layer_backward(L[name], da, P[name], mode, **kw)
        G[name] = g
        if on_grads is not None:
            on_grads(name, g)
        return dx

    d = back("deconv3", dslice(0))
    for i in range(5, -1, -1):
        d = back(f"block3.{i}", d)
    d = back("deconv2", dslice(256), dx=d, dx_accumulate=True)
    for i in range(5, -1, -1):
        d = back(f"block2.{i}", d)
    d = back("deconv1", dslice(512), dx=d, dx_accumulate=True)
    for i in range(4, -1, -1):
        d = back(f"block1.{i}", d)
    d = back("middle_layer.2", d, bev_da=True)
    d = back("middle_layer.1", d)
    if st.sparse is not None:
        g0, d = E.first_layer_backward_sparse(L["middle_layer.0"], d, P["middle_layer.0"], mode, st.sparse[0],
                                              st.sparse[1])
        G["middle_layer.0"] = g0
        if on_grads is not None:
            on_grads("middle_layer.0", g0)
        return G, d          # d: (K,128) fp32 gradient of the voxel features
    d = back("middle_layer.0", d, need_dx=need_dx)
    return G, d
