"""ORACLE — test infrastructure, NOT product code.

PyTorch-CPU fp32 restatement of the floating-point half of the hot path: the
same op sequence as the reference's nn.Modules, written functionally over a
state_dict (stock ATen CPU kernels; torch is un-vendored third-party code, see
SURVEY.md §8c).  References (all under /root/reference/voxelnet/):

  vfe_layer        model.py:60-82     (VFELayer)
  feature_net      model.py:85-108    (FeatureLearningNet)
  conv_md          model.py:111-167   (ConvMD)
  deconv2d         model.py:170-199   (DeConv2d)
  middle_rpn       model.py:202-281   (MiddleConvNet)
  rpn_loss         model.py:341-352 + loss.py:3-13
  make_state_dict  shapes/keys of model.py:284-296

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  Pinned against the imported reference by tests/golden/*.npz
(tools/gen_golden.py), see tests/test_oracle_model.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # torch.nn.BatchNorm* default
BN_MOMENTUM = 0.1   # torch.nn.BatchNorm* default

# (name, dim, cin, cout, k, stride, pad) — model.py:206-254
MIDDLE = [
    ("middle_layer.0", 3, 128, 64, 3, (2, 1, 1), (1, 1, 1)),
    ("middle_layer.1", 3, 64, 64, 3, (1, 1, 1), (0, 1, 1)),
    ("middle_layer.2", 3, 64, 64, 3, (2, 1, 1), (1, 1, 1)),
]


def rpn_layers(cls_name="Car"):
    """Layer table of MiddleConvNet after the Conv3d stack (model.py:212-254)."""
    s1 = (2, 2) if cls_name == "Car" else (1, 1)      # model.py:212-227
    blk1 = [("block1.0", 128, 128, s1)] + [(f"block1.{i}", 128, 128, (1, 1)) for i in range(1, 5)]
    blk2 = [("block2.0", 128, 128, (2, 2))] + [(f"block2.{i}", 128, 128, (1, 1)) for i in range(1, 6)]
    blk3 = [("block3.0", 128, 256, (2, 2))] + [(f"block3.{i}", 256, 256, (1, 1)) for i in range(1, 6)]
    deconvs = [("deconv1", 128, 256, 3, (1, 1), (1, 1)),     # model.py:229
               ("deconv2", 128, 256, 2, (2, 2), (0, 0)),     # model.py:240
               ("deconv3", 256, 256, 4, (4, 4), (0, 0))]     # model.py:251
    return blk1, blk2, blk3, deconvs


def _fill(shape, tag, scale):
    """Closed-form deterministic tensor, uniform in [-scale, scale): splitmix64 of the
    flat index and `tag` (integer arithmetic, identical on every machine).  Random-like
    on purpose: a smooth closed form (e.g. a sine of the index) gives nearly low-rank
    weights, and the 23-layer Conv+BatchNorm+ReLU stack then amplifies fp32 rounding noise
    to O(1) (fp32 and fp64 runs of the SAME torch ops disagree completely) — no
    implementation could be compared against such fixtures."""
    import numpy as np
    n = 1
    for s in shape:
        n *= s
    mask = (1 << 64) - 1
    seed = np.uint64((int(tag) * 0xBF58476D1CE4E5B9 + 0x632BE59BD9B4E019) & mask)
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return torch.from_numpy(((2.0 * u - 1.0) * scale).astype(np.float32)).reshape(shape)


def make_state_dict(cls_name="Car"):
    """state_dict with the reference's keys/shapes and closed-form
    values (no RNG), so fixtures and tests rebuild identical weights."""
    sd = OrderedDict()
    tag = [0]

    def nxt():
        tag[0] += 1
        return tag[0]

    def bn(prefix, c):
        sd[prefix + ".weight"] = 1.0 + _fill((c,), nxt(), 0.2)
        sd[prefix + ".bias"] = _fill((c,), nxt(), 0.1)
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    for name, cin, cout in (("vfe_1", 7, 32), ("vfe_2", 32, 128)):
        c = cout // 2
        sd[f"feature_net.{name}.fcn.0.weight"] = _fill((c, cin), nxt(), 1.0 / math.sqrt(cin))
        sd[f"feature_net.{name}.fcn.0.bias"] = _fill((c,), nxt(), 0.1)
        bn(f"feature_net.{name}.bn", c)
    for name, _dim, cin, cout, k, _s, _p in MIDDLE:
        sd[f"middle_rpn.{name}.conv.weight"] = _fill((cout, cin, k, k, k), nxt(), 1.0 / math.sqrt(cin * k ** 3))
        sd[f"middle_rpn.{name}.conv.bias"] = _fill((cout,), nxt(), 0.1)
        bn(f"middle_rpn.{name}.batch_norm", cout)
    blk1, blk2, blk3, deconvs = rpn_layers(cls_name)

    def convs(block):
        for name, cin, cout, _s in block:
            sd[f"middle_rpn.{name}.conv.weight"] = _fill((cout, cin, 3, 3), nxt(), 1.0 / math.sqrt(cin * 9))
            sd[f"middle_rpn.{name}.conv.bias"] = _fill((cout,), nxt(), 0.1)
            bn(f"middle_rpn.{name}.batch_norm", cout)

    def deconv(d):
        name, cin, cout, k, _s, _p = d
        sd[f"middle_rpn.{name}.deconv.weight"] = _fill((cin, cout, k, k), nxt(), 1.0 / math.sqrt(cin))
        sd[f"middle_rpn.{name}.deconv.bias"] = _fill((cout,), nxt(), 0.1)
        bn(f"middle_rpn.{name}.batch_norm", cout)

    # same registration order as MiddleConvNet.__init__ (model.py:206-254)
    convs(blk1); deconv(deconvs[0]); convs(blk2); deconv(deconvs[1]); convs(blk3); deconv(deconvs[2])
    sd["middle_rpn.prob_conv.conv.weight"] = _fill((2, 768, 1, 1), nxt(), 1.0 / math.sqrt(768))
    sd["middle_rpn.prob_conv.conv.bias"] = _fill((2,), nxt(), 0.1)
    sd["middle_rpn.reg_conv.conv.weight"] = _fill((14, 768, 1, 1), nxt(), 1.0 / math.sqrt(768))
    sd["middle_rpn.reg_conv.conv.bias"] = _fill((14,), nxt(), 0.1)
    return sd


def param_keys(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def _bn(x, sd, prefix, training):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training,
                        BN_MOMENTUM, BN_EPS)


def vfe_layer(inputs, mask, sd, prefix, training=True):
    """model.py:74-82.  inputs (K,T,cin), mask (K,T,1) bool -> (K,T,cout).
    Linear -> ReLU -> BatchNorm1d over channels (stats over all K*T rows, padded
    slots included) -> max over T -> concat -> * mask.  T comes from the tensor."""
    h = F.relu(F.linear(inputs, sd[prefix + ".fcn.0.weight"], sd[prefix + ".fcn.0.bias"]))
    p = _bn(h.transpose(1, 2), sd, prefix + ".bn", training).transpose(1, 2)
    agg = p.max(dim=1, keepdim=True)[0]
    out = torch.cat([p, agg.expand(-1, p.shape[1], -1)], dim=2)
    return out * mask.to(out.dtype)


def voxel_features(feature, sd, training=True):
    """model.py:93-100 without the scatter: (K,T,7) -> (K,128)."""
    mask = feature.max(dim=2, keepdim=True)[0] != 0          # model.py:95-96
    x = vfe_layer(feature, mask, sd, "feature_net.vfe_1", training)
    x = vfe_layer(x, mask, sd, "feature_net.vfe_2", training)
    return x.max(dim=1)[0]                                   # model.py:100


def scatter_dense(voxelwise, coordinate, dims):
    """model.py:102-106: COO (b,z,y,x) rows -> dense (B,D,H,W,C).  Coordinates are
    unique per sample, so index_put == sparse to_dense."""
    B, D, H, W = dims
    dense = voxelwise.new_zeros((B, D, H, W, voxelwise.shape[1]))
    c = coordinate.long()
    return dense.index_put((c[:, 0], c[:, 1], c[:, 2], c[:, 3]), voxelwise)


def feature_net(features, coordinates, sd, dims, training=True):
    """model.py:91-108.  features: list of (K_i,T,7); coordinates: list of (K_i,4)."""
    feature = torch.cat(list(features), dim=0)
    coordinate = torch.cat(list(coordinates), dim=0)
    B = len(features)
    return scatter_dense(voxel_features(feature, sd, training), coordinate, (B,) + tuple(dims))


def _relu(x, relu_mask=None):
    """F.relu, or (tests only) multiplication by a given 0/1 mask: the derivative of ReLU is
    discontinuous at 0, so a checker that must compare BACKWARD passes tightly evaluates the
    reference with the mask of the implementation under test (they differ only where the
    pre-activation is within rounding distance of 0)."""
    return F.relu(x) if relu_mask is None else x * relu_mask.to(x.dtype)


def conv_md(x, sd, prefix, dim, stride, pad, bn=True, act=True, training=True, relu_mask=None):
    """model.py:158-167."""
    fn = F.conv3d if dim == 3 else F.conv2d
    x = fn(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], stride, pad)
    if bn:
        x = _bn(x, sd, prefix + ".batch_norm", training)
    return _relu(x, relu_mask) if act else x


def deconv2d(x, sd, prefix, stride, pad, training=True, relu_mask=None):
    """model.py:195-199."""
    x = F.conv_transpose2d(x, sd[prefix + ".deconv.weight"], sd[prefix + ".deconv.bias"], stride, pad)
    return _relu(_bn(x, sd, prefix + ".batch_norm", training), relu_mask)


def middle_rpn(dense, sd, cls_name="Car", training=True, taps=None, masks=None):
    """model.py:257-281.  dense (B,D,H,W,128) -> (sigmoid(probs) (B,2,h,w), reg (B,14,h,w)).
    `taps`, if a dict, receives intermediate activations for per-layer tests.  `masks`, if a dict name -> 0/1 tensor in
    the layer's NC(D)HW output shape, replaces every F.relu by a multiplication with the given mask (tests only: see _relu)
    — a CHAINED backward can then be compared tightly, with the ReLU decisions of the implementation under test."""
    B, _, H, W, _ = dense.shape
    mk = (lambda n: masks[n]) if masks is not None else (lambda n: None)
    x = dense.permute(0, 4, 1, 2, 3)
    for name, dim, _cin, _cout, _k, s, p in MIDDLE:
        x = conv_md(x, sd, "middle_rpn." + name, dim, s, p, training=training, relu_mask=mk(name))
        if taps is not None:
            taps[name] = x
    x = x.reshape(B, -1, H, W)                               # model.py:262: channel = c*2+d
    blk1, blk2, blk3, dec = rpn_layers(cls_name)
    ups = []
    for block, d in ((blk1, dec[0]), (blk2, dec[1]), (blk3, dec[2])):
        for name, _ci, _co, s in block:
            x = conv_md(x, sd, "middle_rpn." + name, 2, s, (1, 1), training=training, relu_mask=mk(name))
            if taps is not None:
                taps[name] = x
        up = deconv2d(x, sd, "middle_rpn." + d[0], d[4], d[5], training, relu_mask=mk(d[0]))
        if taps is not None:
            taps[d[0]] = up
        ups.append(up)
    x = torch.cat([ups[2], ups[1], ups[0]], dim=1)           # model.py:271-273
    probs = conv_md(x, sd, "middle_rpn.prob_conv", 2, (1, 1), (0, 0), bn=False, act=False)
    reg = conv_md(x, sd, "middle_rpn.reg_conv", 2, (1, 1), (0, 0), bn=False, act=False)
    return torch.sigmoid(probs), reg


def smooth_l1(deltas, targets, sigma=3.0):
    """loss.py:3-13, including its quirk: option1 is multiplied by option2, not by
    the |d|<1/sigma^2 indicator (loss.py:9)."""
    s2 = sigma * sigma
    d = deltas - targets
    sign = (d.abs() < 1.0 / s2).to(d.dtype)
    o1 = d * d * 0.5 * s2
    o2 = d.abs() - 0.5 / s2
    return o1 * o2 + o2 * (1 - sign)


def rpn_loss(prob, delta, pos, neg, targets, alpha=1.5, beta=1.0, sigma=3.0):
    """model.py:310-352 with pos/neg/targets given channels-last (B,h,w,2)/(B,h,w,14)
    float tensors as generate_targets returns them.  -> (loss, cls, reg, cls_pos, cls_neg)."""
    pos_reg = torch.cat([pos[..., [0]].expand(-1, -1, -1, 7), pos[..., [1]].expand(-1, -1, -1, 7)], -1)
    pos_sum = pos.sum(dim=(1, 2, 3)).reshape(-1, 1, 1, 1).clamp(min=1)
    neg_sum = neg.sum(dim=(1, 2, 3)).reshape(-1, 1, 1, 1).clamp(min=1)
    pos_c, neg_c = pos.permute(0, 3, 1, 2), neg.permute(0, 3, 1, 2)
    tgt_c, posr_c = targets.permute(0, 3, 1, 2), pos_reg.permute(0, 3, 1, 2)
    cls_pos = (-pos_c * torch.log(prob + 1e-6)) / pos_sum
    cls_neg = (-neg_c * torch.log(1 - prob + 1e-6)) / neg_sum
    cls = torch.sum(alpha * cls_pos + beta * cls_neg)
    reg = torch.sum(smooth_l1(delta * posr_c, tgt_c * posr_c, sigma) / pos_sum)
    return cls + reg, cls, reg, cls_pos.sum(), cls_neg.sum()


def forward_backward(features, coordinates, sd, dims, cls_name, d_prob, d_reg, training=True, masks=None):
    """One step core (train.py:148-151 minus optimiser) with a supplied upstream gradient: returns (prob, reg,
    {param: grad}).  training=True: train-mode BatchNorm, running stats in `sd` are updated in place like nn.BatchNorm
    does; training=False: the reference's autograd through `model.eval()` (running statistics as constants)."""
    keys = param_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    dense = feature_net(features, coordinates, work, dims, training)
    prob, reg = middle_rpn(dense, work, cls_name, training, masks=masks)
    torch.autograd.backward([prob, reg], [d_prob, d_reg])
    return prob.detach(), reg.detach(), {k: leaves[k].grad for k in keys}


def clip_sgd_step(params, grads, lr=0.01, max_norm=5.0):
    """train.py:153-154 with the optimizer of train.py:130 restated: clip_grad_norm_(params, max_norm) — 2-norm over
    all gradients together, coef = max_norm / (total + 1e-6) clamped to 1 — then SGD(lr) without momentum or weight
    decay.  -> (new params, clipped grads, total norm); computes in the tensors' dtype."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    clipped = [g * coef for g in grads]
    return [p - lr * g for p, g in zip(params, clipped)], clipped, total


def train_step(features, coordinates, sd, dims, cls_name, targets, lr=0.01, max_norm=5.0, alpha=1.5, beta=1.0, sigma=3.0):
    """One iteration of the reference's train loop (train.py:148-155: forward with the loss of model.py:310-352,
    loss.backward(), clip_grad_norm_(params, max_norm), SGD(lr).step(), zero_grad()) on the state dict `sd`, IN PLACE:
    parameters updated, BatchNorm running statistics updated by the forward, num_batches_tracked += 1.
    targets = (pos_equal_one (B,h,w,2), neg_equal_one (B,h,w,2), targets (B,h,w,14)) float tensors, what
    utils.generate_targets returns at model.py:309.  -> ([loss, cls, reg, cls_pos, cls_neg] floats, total gradient norm)."""
    keys = param_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    dense = feature_net(features, coordinates, work, dims, True)
    prob, reg = middle_rpn(dense, work, cls_name, True)
    out = rpn_loss(prob, reg, *targets, alpha, beta, sigma)
    out[0].backward()
    new, _, total = clip_sgd_step([leaves[k].detach() for k in keys], [leaves[k].grad for k in keys], lr, max_norm)
    for k, v in zip(keys, new):
        sd[k] = v
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = sd[k] + 1
    return [float(v) for v in out], float(total)
