"""Optimizer tail of the reference's train step (voxelnet/train.py:153-154):

    clip_grad_norm_(model.parameters(), 5)        # train.py:153
    optimizer.step()                              # train.py:154, optimizer = SGD(model.parameters(), lr=0.01)

as two HIP launches (csrc/optim.hip, `vn_clip_sgd`) over a device chunk table of the (parameter, gradient) pairs,
whatever their placement (the module's flat gradient buffer, the DDP buckets' views, or 104 separate tensors).
No CPU / torch fallback: the HIP library must be present."""
import ctypes

import numpy as np
import torch

from . import _lib

CHUNK = 4096    # VN_OPT_CHUNK (include/voxelnet_hip.h)


class ClipSGD(torch.optim.Optimizer):
    """`ClipSGD(params, lr, max_norm).step()` == `clip_grad_norm_(params, max_norm); SGD(params, lr).step()`.
    step() returns the total gradient norm before clipping (a device scalar, clip_grad_norm_'s return value).

    A torch.optim.Optimizer: `param_groups` / `state_dict()` / `load_state_dict()` / `zero_grad()` are the base class's, so
    the reference's `MultiStepLR(optimizer, ...)` (train.py:131) attaches and its lr changes are honoured: the learning
    rate is read from `param_groups[0]["lr"]` at every step.  The norm is taken over ALL parameters together (as
    train.py:153 does), so every group must carry the same lr / max_norm."""

    def __init__(self, params, lr, max_norm, scale_grads=False):
        defaults = dict(lr=float(lr), max_norm=float(max_norm), scale_grads=bool(scale_grads))
        super().__init__(params, defaults)
        if not any(len(g["params"]) for g in self.param_groups):
            raise ValueError("ClipSGD got an empty parameter list")
        self._key = None
        self._table = self._ws = self._norm = None
        self._n_chunks = 0
        self._plist = self._last_grads = self._last_pptrs = self._fused_key = None

    # (kept for callers of the round-1 class)
    @property
    def params(self):
        return [p for g in self.param_groups for p in g["params"]]

    @property
    def lr(self):
        return float(self.param_groups[0]["lr"])

    @property
    def max_norm(self):
        return float(self.param_groups[0]["max_norm"])

    @property
    def scale_grads(self):
        return bool(self.param_groups[0]["scale_grads"])

    def __setstate__(self, state):        # (the base class pickles defaults / state / param_groups only)
        super().__setstate__(state)
        self._key = None                  # device chunk table / workspace: rebuilt on the first step
        self._table = self._ws = self._norm = None
        self._n_chunks = 0
        self._plist = self._last_grads = self._last_pptrs = self._fused_key = None

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad without its per-parameter foreach bookkeeping (104 small tensors)"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for p in self.params:
            p.grad = None

    def add_param_group(self, group):
        super().add_param_group(group)
        self._plist = self._last_grads = self._last_pptrs = self._fused_key = None

    def _build(self, pairs, dev):
        rows = []
        for p, g in pairs:
            n, pp, gp = p.numel(), p.data_ptr(), g.data_ptr()
            for off in range(0, n, CHUNK):
                rows.append((pp + 4 * off, gp + 4 * off, min(CHUNK, n - off), 0))
        tab = np.array(rows, dtype=np.dtype([("param", "<u8"), ("grad", "<u8"), ("n", "<i4"), ("reserved", "<i4")]))
        assert tab.dtype.itemsize == ctypes.sizeof(_lib.VnParamChunk)
        self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        self._n_chunks = len(rows)
        nbytes = _lib.load().vn_clip_sgd_workspace_bytes(self._n_chunks)
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def _step_table(self, params, grads):
        """RPN3D.train_step (vn_net_step): make sure the device chunk table covers exactly the pairs (params[i], grads[i]) —
        the gradients are not attached to the parameters yet, the update runs inside the library call — and that this
        optimizer's parameters are those.  -> True: _table / _ws / _norm are valid for the call; False: use step()."""
        for g in self.param_groups[1:]:
            if g["lr"] != self.param_groups[0]["lr"] or g["max_norm"] != self.param_groups[0]["max_norm"]:
                return False
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self._plist = self.params
        pptrs = [p.data_ptr() for p in params]
        hit = self.__dict__.get("_fused_key")
        if hit is not None and hit[0] is params and hit[1] is grads and hit[2] == pptrs and self._table is not None:
            return True
        if len(plist) != len(params) or {id(p) for p in plist} != {id(p) for p in params}:
            return False
        dev = params[0].device
        for p, g in zip(params, grads):
            if not (p.is_cuda and g.is_cuda and p.device == dev and g.device == dev and p.dtype == torch.float32
                    and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and p.numel() == g.numel()):
                return False
        gmap = {id(p): g for p, g in zip(params, grads)}
        pairs = [(p, gmap[id(p)]) for p in plist]
        key = tuple((p.data_ptr(), g.data_ptr(), p.numel()) for p, g in pairs)
        if key != self._key:
            self._build(pairs, dev)
            self._key = key
        self._n_elems = sum(p.numel() for p in plist)
        self._last_grads = [gmap[id(p)] for p in plist]      # (step() right after the call would see the same tensors)
        self._last_pptrs = [p.data_ptr() for p in plist]
        self._fused_key = (params, grads, pptrs)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise _lib.VoxelnetHipError("ClipSGD.step: closures are not supported")
        for g in self.param_groups[1:]:
            if g["lr"] != self.param_groups[0]["lr"] or g["max_norm"] != self.param_groups[0]["max_norm"]:
                raise _lib.VoxelnetHipError("ClipSGD: one lr / max_norm for all parameter groups (the clip norm is global)")
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self._plist = self.params
        last = self.__dict__.get("_last_grads")
        if (last is not None and self._table is not None and all(p.grad is g for p, g in zip(plist, last))
                and self._last_pptrs == [p.data_ptr() for p in plist]):
            # the same gradient tensors as in the previous step (the model's flat buffer / bucket views) and the same
            # parameter storage (a `p.data = ...` / `set_()` swap keeps the Parameter object but not its memory): the chunk
            # table of raw pointers is still valid, nothing to rebuild or re-check
            with _lib.on_device(self._table.device):
                stream = _lib.raw_stream()
                from . import engine as E
                with E.section("clip_sgd", 16.0 * self._n_elems):
                    _lib.call("vn_clip_sgd", self._table.data_ptr(), self._n_chunks, self.max_norm, self.lr,
                              int(self.scale_grads), self._ws.data_ptr(), self._ws.numel(), self._norm.data_ptr(), stream)
            return self._norm[0]
        pairs = [(p, p.grad) for p in plist if p.grad is not None]
        if not pairs:
            return None
        dev = pairs[0][0].device
        for p, g in pairs:
            if not (p.is_cuda and g.is_cuda and p.device == dev and g.device == dev):
                raise _lib.VoxelnetHipError("ClipSGD: parameters and gradients must live on one HIP device (no CPU path)")
            if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous():
                raise _lib.VoxelnetHipError("ClipSGD: fp32 contiguous parameters and gradients only")
        key = tuple((p.data_ptr(), g.data_ptr(), p.numel()) for p, g in pairs)
        if key != self._key:
            self._build(pairs, dev)        # pointers are stable from step to step (flat gradient buffer): built once
            self._key = key
        self._n_elems = sum(p.numel() for p, _ in pairs)
        self._last_grads = [p.grad for p in plist] if len(pairs) == len(plist) else None
        self._last_pptrs = [p.data_ptr() for p in plist]
        with _lib.on_device(dev):
            stream = _lib.raw_stream()
            from . import engine as E
            with E.section("clip_sgd", 16.0 * sum(p.numel() for p, _ in pairs)):      # grad read twice, param read + written
                _lib.call("vn_clip_sgd", self._table.data_ptr(), self._n_chunks, self.max_norm, self.lr, int(self.scale_grads),
                          self._ws.data_ptr(), self._ws.numel(), self._norm.data_ptr(), stream)
        return self._norm[0]
