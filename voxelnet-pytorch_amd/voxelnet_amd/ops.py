"""torch.autograd.Function wrappers over the C ABI (include/voxelnet_hip.h).
PyTorch supplies device memory, the current HIP stream and the autograd tape;
all arithmetic happens in libvoxelnet_hip.so."""
import ctypes

import torch

from . import _lib
from ._lib import VN_BF16, VN_F32


def stream():
    return _lib.raw_stream()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VoxelnetHipError("voxelnet_amd ops need CUDA(HIP) tensors; there is no CPU path")


def vn_dtype(t):
    if t.dtype == torch.float32:
        return VN_F32
    if t.dtype == torch.bfloat16:
        return VN_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


class _ScatterDense(torch.autograd.Function):
    """model.py:102-106 (sparse COO -> dense) / backward = row gather."""

    @staticmethod
    def forward(ctx, voxelwise, coord, dims, out_dtype, split3):
        _need_cuda(voxelwise, coord)
        B, D, H, W = dims
        K, C = voxelwise.shape
        voxelwise = voxelwise.contiguous().float()
        coord = coord.contiguous()
        ch = 3 * C if split3 else C
        dense = torch.empty((B, D, H, W, ch), dtype=out_dtype, device=voxelwise.device)
        with _lib.on_device(voxelwise.device):
            _lib.call("vn_scatter_dense_fwd", voxelwise.data_ptr(), coord.data_ptr(), K, C, B, D, H, W,
                      dense.data_ptr(), vn_dtype(dense), ch, int(split3), stream())
        ctx.save_for_backward(coord)
        ctx.meta = (K, C, B, D, H, W)
        return dense

    @staticmethod
    def backward(ctx, d_dense):
        (coord,) = ctx.saved_tensors
        K, C, B, D, H, W = ctx.meta
        d_dense = d_dense.contiguous()
        if d_dense.shape[-1] != C:
            raise _lib.VoxelnetHipError("gradient of a split3 grid must be folded to C channels first")
        d_vw = torch.empty((K, C), dtype=torch.float32, device=d_dense.device)
        with _lib.on_device(d_dense.device):
            _lib.call("vn_scatter_dense_bwd", d_dense.data_ptr(), vn_dtype(d_dense), coord.data_ptr(), K, C, B, D,
                      H, W, d_vw.data_ptr(), stream())
        return d_vw, None, None, None, None


def scatter_dense(voxelwise, coord, dims, out_dtype=torch.float32, split3=False):
    return _ScatterDense.apply(voxelwise, coord, tuple(dims), out_dtype, split3)
