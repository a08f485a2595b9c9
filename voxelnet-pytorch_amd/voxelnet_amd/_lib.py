"""ctypes binding of libvoxelnet_hip.so (C ABI: include/voxelnet_hip.h).

There is NO fallback: if the library is missing, or a call returns a non-zero
status, this raises.  PyTorch is used above this layer only for device memory,
streams and autograd bookkeeping.
"""
import ctypes
import os

# torch must be imported BEFORE the library is dlopen'ed: both need
# libamdhip64.so.7 and must share ONE HIP runtime instance (the one bundled with
# torch).  Loaded in the other order the process ends up with two runtimes and
# our launches fail with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# VN_LIB_PATH: another build of the library (tools/ab_bench.sh: interleaved A/B of two builds inside one gpurun call)
LIB_PATH = os.environ.get("VN_LIB_PATH") or os.path.join(_HERE, "lib", "libvoxelnet_hip.so")

c_i32, c_i64, c_f32, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

VN_F32, VN_BF16 = 0, 1


class VnGrid(ctypes.Structure):
    _fields_ = [("D", c_i32), ("H", c_i32), ("W", c_i32),
                ("vz", c_f32), ("vy", c_f32), ("vx", c_f32),
                ("ox", c_f32), ("oy", c_f32), ("oz", c_f32), ("T", c_i32)]


class VnConv(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in (
        "dtype", "B", "Ds", "Hs", "Ws", "Dr", "Hr", "Wr", "Cs", "src_wrap", "Cr", "kD", "kH", "kW",
        "mulD", "mulH", "mulW", "tmulD", "tmulH", "tmulW", "padD", "padH", "padW",
        "divD", "divH", "divW")] + [(n, c_i64) for n in (
        "src_sB", "src_sD", "src_sH", "src_sW", "out_sB", "out_sD", "out_sH", "out_sW")]


class VnVfeWeights(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("w1", "b1", "g1", "be1", "rm1", "rv1", "w2", "b2", "g2", "be2", "rm2", "rv2")]


class VnVfeGrads(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("dw1", "db1", "dg1", "dbe1", "dw2", "db2", "dg2", "dbe2")]


class VnNetConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("B", "D", "H", "W", "block1_stride", "mode", "training", "sparse_first", "prepared", "bucket_events", "defer_join", "grad_storage")]


class VnTimingRecord(ctypes.Structure):
    _fields_ = [("kind", c_i32), ("layer", c_i32), ("ms", c_f32), ("start_ms", c_f32), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


class VnLayerParams(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("weight", "bias", "gamma", "beta", "running_mean", "running_var")]


class VnLayerGrads(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("weight", "bias", "gamma", "beta")]


class VnPackJob(ctypes.Structure):
    _fields_ = [("w", c_vp), ("packed", c_vp)] + [(n, c_i32) for n in ("c_out", "c_in", "taps", "mode", "split3", "cin_fold",
                                                                       "packed_dtype", "pad_")]


class VnUnpackJob(ctypes.Structure):
    _fields_ = [("dw_packed", c_vp), ("dw", c_vp)] + [(n, c_i32) for n in ("c_out", "c_in", "taps", "mode", "cin_fold", "chunks")] + \
               [("chunk_stride", c_i64)]


class VnParamChunk(ctypes.Structure):
    _fields_ = [("param", c_vp), ("grad", c_vp), ("n", c_i32), ("reserved", c_i32)]


class VnStep(ctypes.Structure):       # vnStep (vn_net_step): field for field
    _fields_ = [("feature", c_vp), ("coord", c_vp), ("K", c_i64), ("T", c_i32), ("bn_momentum", c_f32), ("bn_eps", c_f32),
                ("vfe", VnVfeWeights), ("vfe_grads", VnVfeGrads), ("vfe_ws", c_vp), ("vfe_ws_bytes", c_sz),
                ("voxelwise", c_vp), ("vfe_stats", c_vp), ("vw_rows", c_vp), ("d_voxelwise", c_vp),
                ("prob_w", c_vp), ("prob_b", c_vp), ("reg_w", c_vp), ("reg_b", c_vp), ("heads_w", c_vp), ("heads_b", c_vp),
                ("d_heads_w", c_vp), ("d_heads_b", c_vp), ("layers", c_vp), ("grads", c_vp), ("ws", c_vp), ("ws_bytes", c_sz),
                ("prob", c_vp), ("reg", c_vp), ("d_prob", c_vp), ("d_reg", c_vp), ("pos", c_vp), ("neg", c_vp),
                ("targets", c_vp), ("targets_stream", c_vp), ("alpha", c_f32), ("beta", c_f32), ("sigma", c_f32),
                ("loss_ws", c_vp), ("loss_ws_bytes", c_sz), ("loss5", c_vp), ("g_loss", c_vp), ("chunks", c_vp),
                ("n_chunks", c_i32), ("max_norm", c_f32), ("lr", c_f32), ("scale_grads", c_i32), ("opt_ws", c_vp),
                ("opt_ws_bytes", c_sz), ("total_norm", c_vp), ("bn_counters", c_vp), ("n_bn_counters", c_i32), ("stream", c_vp),
                ("side_stream", c_vp)]


# name -> (restype, argtypes); mirrors include/voxelnet_hip.h one to one
_P = ctypes.POINTER
SIGNATURES = {
    "vn_abi_version": (c_i32, []),
    "vn_build_info": (ctypes.c_char_p, []),
    "vn_build_id": (ctypes.c_char_p, []),
    "vn_voxelize_workspace_bytes": (c_sz, [c_i64, _P(VnGrid)]),
    "vn_voxelize_index": (c_i32, [c_vp, c_i64, _P(VnGrid), c_vp, c_sz, c_vp, c_vp]),
    "vn_voxelize_gather": (c_i32, [c_vp, c_i64, _P(VnGrid), c_vp, c_sz, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vn_voxelize_host_workspace_bytes": (c_sz, [c_i64, _P(VnGrid)]),
    "vn_voxelize_host_index": (c_i32, [c_vp, c_i64, _P(VnGrid), c_vp, c_sz, _P(c_i64)]),
    "vn_voxelize_host_gather": (c_i32, [c_vp, c_i64, _P(VnGrid), c_vp, c_sz, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vn_vfe_layer_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32, c_i32]),
    "vn_vfe_layer_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32,
                                 c_vp, c_vp, c_sz, c_vp]),
    "vn_vfe_layer_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_sz, c_vp]),
    "vn_fov_crop_workspace_bytes": (c_sz, [c_i64]),
    "vn_fov_crop": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "vn_comm_rccl_version": (c_i32, []),
    "vn_comm_unique_id": (c_i32, [c_vp]),
    "vn_comm_create": (c_i32, [_P(c_vp), c_vp, c_i32, c_i32]),
    "vn_comm_destroy": (c_i32, [c_vp]),
    "vn_allreduce_bucket": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "vn_vfe_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "vn_vfe_fwd": (c_i32, [c_vp, c_i64, c_i32, _P(VnVfeWeights), c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "vn_vfe_fwd_rows": (c_i32, [c_vp, c_i64, c_i32, _P(VnVfeWeights), c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "vn_vfe_bwd": (c_i32, [c_vp, c_i64, c_i32, _P(VnVfeWeights), c_vp, c_vp, _P(VnVfeGrads), c_vp, c_sz, c_i32, c_vp]),
    "vn_scatter_dense_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32,
                                     c_i32, c_vp]),
    "vn_scatter_dense_update": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32,
                                        c_i32, c_vp]),
    "vn_scatter_dense_bwd": (c_i32, [c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vn_conv_gather_gemm": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, _P(VnConv), c_i32, c_vp, c_vp]),
    "vn_net_workspace_bytes": (c_sz, [_P(VnNetConfig), c_i64]),
    "vn_net_create": (c_i32, [_P(c_vp)]),
    "vn_net_destroy": (c_i32, [c_vp]),
    "vn_net_step": (c_i32, [c_vp, _P(VnNetConfig), _P(VnStep)]),
    "vn_net_timing_begin": (c_i32, [c_vp, c_i32]),
    "vn_net_timing_read": (c_i32, [c_vp, c_vp, c_i32, _P(c_i32)]),
    "vn_net_forward": (c_i32, [c_vp, _P(VnNetConfig), _P(VnLayerParams), c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp,
                               c_vp, c_vp, c_vp]),
    "vn_net_wait_bucket": (c_i32, [c_vp, c_i32, c_vp]),
    "vn_net_prepare": (c_i32, [c_vp, _P(VnNetConfig), _P(VnLayerParams), c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "vn_voxel_index_grid": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vn_rulebook_slab_rows": (c_i64, [c_i64]),
    "vn_rulebook_combine": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, _P(VnConv), c_vp, c_vp, c_i32, c_vp, c_vp]),
    "vn_net_backward": (c_i32, [c_vp, _P(VnNetConfig), _P(VnLayerParams), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                c_vp, c_sz, _P(VnLayerGrads), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "vn_conv_stats_slab_rows": (c_i64, [_P(VnConv)]),
    "vn_conv_plan_id": (c_i32, [_P(VnConv)]),
    "vn_conv_wgrad_plan_id": (c_i32, [_P(VnConv), c_i32, c_i64]),
    "vn_bn_finalize_slab": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp]),
    "vn_conv_wgrad_workspace_bytes": (c_sz, [_P(VnConv), c_i32, c_i64]),
    "vn_conv_wgrad": (c_i32, [c_vp, c_vp, c_vp, _P(VnConv), c_i32, c_vp, c_sz, c_vp]),
    "vn_conv_dgrad_bn_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, _P(VnConv), c_vp, c_i32, c_vp, c_vp, c_vp]),
    "vn_conv_gather_gemm_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, _P(VnConv), c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "vn_conv_wgrad_partials": (c_i32, [c_vp, c_vp, _P(VnConv), c_i32, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp]),
    "vn_conv_wgrad_rows": (c_i32, [c_vp, c_vp, c_vp, _P(VnConv), c_vp, c_i64, c_vp, c_sz, c_vp]),
    "vn_conv_wgrad_partials_split_pass": (c_i32, [c_vp, c_vp, _P(VnConv), c_i32, c_vp, c_sz, c_vp, c_vp]),
    "vn_active_sites_workspace_bytes": (c_sz, [_P(VnConv)]),
    "vn_active_sites": (c_i32, [c_vp, c_i64, _P(VnConv), c_vp, c_sz, c_vp, c_i64, c_vp, c_vp]),
    "vn_fill_rows": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_vp]),
    "vn_pack_weight": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "vn_unpack_wgrad": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vn_bn_stats": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vn_bn_finalize": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_vp,
                               c_vp]),
    "vn_bn_apply": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_vp]),
    "vn_bn_bwd_reduce": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "vn_bn_bwd_finalize": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vn_bn_bwd_slab_rows": (c_i64, [c_i64, c_i32]),
    "vn_bn_bwd_reduce_slab": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "vn_bn_bwd_finalize_slab": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vn_bn_bwd_apply": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp,
                                c_i32, c_i64, c_i64, c_vp]),
    "vn_bn_apply_bev": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_i32, c_i64, c_vp]),
    "vn_bn_bwd_reduce_slab_bev": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "vn_bn_bwd_apply_bev": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "vn_bn_apply_flagged": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "vn_bn_bwd_reduce_slab_flagged": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp,
                                              c_vp]),
    "vn_bn_bwd_apply_list": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp,
                                     c_vp, c_i64, c_vp]),
    "vn_bn_bwd_list_slab_rows": (c_i64, [c_i64, c_i32]),
    "vn_bn_bwd_reduce_list": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vn_bn_bwd_finalize_list": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "vn_bn_bwd_apply_list_rows": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32,
                                          c_vp, c_vp, c_i64, c_vp]),
    "vn_dgrad_total_workspace_bytes": (c_sz, [c_i32]),
    "vn_dgrad_total": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_sz, c_vp, c_vp]),
    "vn_box_col_sums": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "vn_act_delta_rows": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i32,
                                  c_vp]),
    "vn_wgrad_const_add": (c_i32, [c_vp, c_vp, c_sz, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vn_conv_wgrad_partials_counted": (c_i32, [c_vp, c_vp, _P(VnConv), c_vp, c_i64, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "vn_bn_bwd_apply_flagged": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32,
                                        c_i64, c_vp, c_vp]),
    "vn_nchw_to_rows": (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp, c_i32, c_i64, c_i64, c_vp]),
    "vn_rows_to_nchw": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i64, c_vp, c_i32, c_vp]),
    "vn_cast_rows": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_i32, c_i64, c_i64, c_vp]),
    "vn_col_sums_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "vn_col_sums": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "vn_heads_to_nchw": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "vn_heads_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "vn_heads_dgrad": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "vn_heads_dgrad_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i64, c_i64, c_vp]),
    "vn_heads_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vn_pack_weights_batch": (c_i32, [c_vp, c_i32, c_vp]),
    "vn_unpack_wgrads_batch": (c_i32, [c_vp, c_i32, c_vp]),
    "vn_rpn_loss_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "vn_rpn_loss_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_sz, c_vp, c_vp]),
    "vn_rpn_loss_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vn_rpn_loss_norm": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "vn_rpn_loss_fwd_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_sz, c_vp, c_vp,
                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vn_rpn_loss_finalize": (c_i32, [c_vp, c_sz, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp]),
    "vn_rpn_loss_fwd_bwd_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_sz, c_vp, c_vp,
                                         c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "vn_rpn_targets_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "vn_rpn_targets": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, ctypes.c_double, c_vp, c_vp, c_vp,
                               c_vp, c_sz, c_vp]),
    "vn_rpn_predict_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "vn_rpn_predict": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, ctypes.c_double, c_i32, ctypes.c_double, c_vp, c_vp, c_vp,
                               c_vp, c_sz, c_vp]),
    "vn_clip_sgd_workspace_bytes": (c_sz, [c_i32]),
    "vn_clip_sgd": (c_i32, [c_vp, c_i32, c_f32, c_f32, c_i32, c_vp, c_sz, c_vp, c_vp]),
}

_lib = None


class VoxelnetHipError(RuntimeError):
    pass


ABI_VERSION = 4     # include/voxelnet_hip.h: vn_abi_version()


def load():
    """Load the HIP library; loud failure if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoxelnetHipError(
                f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C voxelnet-pytorch_amd/csrc).  There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        have = lib.vn_abi_version() if hasattr(lib, "vn_abi_version") else None
        if have != ABI_VERSION:
            raise VoxelnetHipError(f"{LIB_PATH} has ABI version {have}, this package binds version {ABI_VERSION}: "
                                   "rebuild it (make -C voxelnet-pytorch_amd/csrc)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def missing_symbols():
    """Symbols declared in include/voxelnet_hip.h that the built library lacks
    (must be empty; __graft_entry__.build() and tests/test_abi.py assert it)."""
    lib = load()
    return [n for n in SIGNATURES if not hasattr(lib, n)]


def check(status, what):
    if status != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "workspace too small"}.get(
            status, f"hipError_t {status}" if status > 0 else "error")
        raise VoxelnetHipError(f"{what} failed: {kind} (status {status})")


def call(name, *args):
    fn = getattr(load(), name, None)
    if fn is None:
        raise VoxelnetHipError(f"{name} is not exported by {LIB_PATH}; rebuild the HIP library")
    check(fn(*args), name)


def raw_stream():
    """hipStream_t of torch's current stream on the current device as the C ABI takes it (vnStream).  The two private
    accessors cost ~0.2 us; torch.cuda.current_stream().cuda_stream builds a Stream object through three Python layers
    (~9 us, 24 times per train step: 0.2 ms of the host's step time, tools/host_cprofile.py)."""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _Here:
    """no-op context: the wanted device is already the current one"""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def on_device(device):
    """`with on_device(t.device):` — torch.cuda.device(device) only when it is not the current device already (the usual
    case: one process per GPU).  The real context manager costs ~10 us of Python per use, ~15 uses per train step."""
    import torch
    idx = getattr(device, "index", device)
    if idx is None or idx == torch._C._cuda_getDevice():
        return _HERE
    return torch.cuda.device(device)


_STREAM_OBJS = {}


def current_stream(device=None):
    """torch.cuda.current_stream(device) without building a new Stream object per call (~6 us through three Python layers):
    the (stream id, device index, device type) triple of torch's C accessor keys a cache of Stream objects"""
    import torch
    idx = getattr(device, "index", device)
    if idx is None:
        idx = torch._C._cuda_getDevice()
    key = torch._C._cuda_getCurrentStream(idx)
    st = _STREAM_OBJS.get(key)
    if st is None:
        st = _STREAM_OBJS[key] = torch.cuda.current_stream(idx)
    return st
