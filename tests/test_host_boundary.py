"""CPU (no GPU needed): the parts of the drop-in boundary that live on the host.

  * vn_voxelize_host_index / _gather — the entry the reference's own call site needs (pcl_to_voxels inside forked
    DataLoader workers, dataset.py:58 / train.py:77-84): bit-exact against the golden vectors of the imported reference
    (utils.py:10-100) and against the oracle, including a real DataLoader with worker processes.
  * torch.save(model) / copy.deepcopy(model) with the executor's runtime caches present (train.py:24/27 pickles the
    whole module), ClipSGD as a torch.optim.Optimizer under MultiStepLR (train.py:130-132)."""
import copy
import ctypes
import hashlib
import io

import numpy as np
import pytest
import torch

from oracle import voxelize as ov
from voxelnet_amd import synth
from voxelnet_amd.config import grid_config


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("tag,target", [("car", "Car"), ("ped", "Pedestrian")])
def test_host_voxelizer_bit_exact_vs_reference_golden(golden, tag, target):
    from voxelnet_amd.voxelize import voxelize_host
    g = golden(f"voxelize_{tag}_small")
    with np.errstate(all="ignore"):
        f, c, n = voxelize_host(g["points"], grid_config(target), 0, coord_cols=3)
    assert c.dtype == np.int64 and np.array_equal(c, g["coordinate_buffer"])
    assert np.array_equal(n, g["number_buffer"])
    assert f.dtype == np.float32 and np.array_equal(f.view(np.uint32), g["feature_buffer"].view(np.uint32))


def test_host_voxelizer_degenerate_and_digest(golden):
    from voxelnet_amd.voxelize import voxelize_host
    g = golden("voxelize_degenerate")
    f, c, n = voxelize_host(g["far_points"], grid_config("Car"), 0, 3)
    assert f.shape == (0, 35, 7) and c.shape == (0, 3) and n.shape == (0,)
    f, c, n = voxelize_host(g["one_points"], grid_config("Car"), 0, 3)
    assert np.array_equal(f, g["one_feature"]) and np.array_equal(c, g["one_coord"]) and np.array_equal(n, g["one_number"])
    f, c, n = voxelize_host(np.zeros((0, 4), np.float32), grid_config("Car"), 0, 3)
    assert n.shape == (0,)
    d = golden("voxelize_full_digest")
    for cfg_id, target in ((2, "Car"), (3, "Pedestrian")):
        w = synth.WORKLOADS[cfg_id]
        cloud = synth.synth_cloud(target, w["k0"], synth.frame_seed(cfg_id, 0), w["mean_extra"], w["T"])
        np.random.seed(7)
        np.random.shuffle(cloud)
        f, c, n = voxelize_host(cloud, grid_config(target), 0, 3)
        assert sha(c) == str(d[f"cfg{cfg_id}_coord_sha"]) and sha(n) == str(d[f"cfg{cfg_id}_number_sha"])
        assert sha(f) == str(d[f"cfg{cfg_id}_feature_sha"])


def test_host_voxelizer_dense_T64_and_batch_column():
    """BASELINE configs[4] shape class (T = 64, ~300k points) against the oracle; coord_cols = 4 adds the batch index"""
    from voxelnet_amd.voxelize import voxelize_host
    w = synth.WORKLOADS[5]
    cloud = synth.synth_cloud(w["target"], 8000, 123, w["mean_extra"], w["T"])
    grid = grid_config("Car", T=64)
    f, c, n = voxelize_host(cloud, grid, 3, coord_cols=4)
    ref = ov.voxelize(cloud, "Car", T=64)
    assert np.array_equal(c[:, 1:], ref["coordinate_buffer"]) and (c[:, 0] == 3).all()
    assert np.array_equal(n, ref["number_buffer"]) and int(n.max()) == 64
    assert np.array_equal(f.view(np.uint32), ref["feature_buffer"].view(np.uint32))


def test_host_entry_status_codes():
    from voxelnet_amd import _lib
    lib = _lib.load()
    g = _lib.VnGrid(10, 400, 352, 0.4, 0.2, 0.2, 0.0, 40.0, 3.0, 35)
    bad = _lib.VnGrid(10, 400, 352, 0.4, 0.2, 0.2, 0.0, 40.0, 3.0, 99)
    assert lib.vn_voxelize_host_workspace_bytes(1000, ctypes.byref(bad)) == 0
    nbytes = lib.vn_voxelize_host_workspace_bytes(1000, ctypes.byref(g))
    ws = np.empty(nbytes, np.uint8)
    pts = np.zeros((1000, 4), np.float32)
    k = ctypes.c_int64(-1)
    assert lib.vn_voxelize_host_index(pts.ctypes.data, 1000, ctypes.byref(g), ws.ctypes.data, 16, ctypes.byref(k)) == -3
    assert lib.vn_voxelize_host_index(pts.ctypes.data, 1000, ctypes.byref(g), None, nbytes, ctypes.byref(k)) == -1
    assert lib.vn_voxelize_host_index(pts.ctypes.data, 1000, ctypes.byref(g), ws.ctypes.data, nbytes, ctypes.byref(k)) == 0
    assert k.value == 1      # all points at the origin: one voxel
    # a K that is not the index phase's is refused
    f = np.empty((2, 35, 7), np.float32); c = np.empty((2, 3), np.int64); n = np.empty(2, np.int64)
    assert lib.vn_voxelize_host_gather(pts.ctypes.data, 1000, ctypes.byref(g), ws.ctypes.data, nbytes, 2, 0, 3, f.ctypes.data,
                                       c.ctypes.data, n.ctypes.data) == -1


class _Clouds(torch.utils.data.Dataset):
    """what KITTIDataset.__getitem__ does at dataset.py:58: voxelize inside the worker"""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from voxelnet_amd.voxelize import pcl_to_voxels
        cloud = synth.synth_cloud("Car", 300, 40 + i)
        np.random.seed(100 + i)
        shuffled = cloud.copy()
        v = pcl_to_voxels(shuffled, "Car")          # default device: worker process -> the host entry
        return str(i), None, shuffled, [], v


def test_pcl_to_voxels_runs_in_dataloader_workers():
    """the reference's data path (train.py:77-84): DataLoader(num_workers > 0, collate_fn) with the voxelizer in the workers"""
    from voxelnet_amd.voxelize import collate_fn
    loader = torch.utils.data.DataLoader(_Clouds(4), batch_size=2, num_workers=2, collate_fn=collate_fn)
    seen = 0
    for tag, label, feats, numbers, coords, rgb, lidar in loader:
        for b in range(len(tag)):
            ref = ov.voxelize(lidar[b], "Car")      # lidar[b] is the shuffled cloud the worker voxelized
            assert np.array_equal(feats[b].numpy().view(np.uint32), ref["feature_buffer"].view(np.uint32))
            assert np.array_equal(coords[b].numpy()[:, 1:], ref["coordinate_buffer"])
            assert (coords[b].numpy()[:, 0] == b).all()
            seen += 1
    assert seen == 4


def test_model_pickles_and_deepcopies_with_runtime_caches():
    """torch.save(model) is the reference's checkpoint format (train.py:24, 27); the executor's caches must not travel"""
    from voxelnet_amd import _lib
    from voxelnet_amd import model as M
    m = M.RPN3D("Car")
    # what a few steps leave in the instance dicts (a GPU is not needed to create them)
    arr = (_lib.VnLayerParams * 23)()
    m.middle_rpn.__dict__["_native_arrays"] = {False: (("key",), arr, None)}
    m.__dict__["_ws_pool"] = [torch.empty(1 << 20, dtype=torch.uint8)]
    m.__dict__["_flat_grads"] = (("k",), torch.zeros(8), {"x": torch.zeros(8)})
    m.__dict__["_net_ctx"] = None
    m.__dict__["_flat_param_list"] = list(m.parameters())
    buf = io.BytesIO()
    torch.save(m, buf)
    assert buf.tell() < 40 << 20           # 27 MB of parameters, not the arenas
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert "_ws_pool" not in m2.__dict__ and "_native_arrays" not in m2.middle_rpn.__dict__
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
    m3 = copy.deepcopy(m)
    assert "_flat_grads" not in m3.__dict__ and m3.grad_reducer is None
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    assert "_ws_pool" in m.__dict__        # the original keeps its caches


def test_clip_sgd_is_a_torch_optimizer():
    from voxelnet_amd.optim import ClipSGD
    ps = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(3, 3))]
    opt = ClipSGD(ps, 0.01, 5.0)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]["lr"] == 0.01
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)     # train.py:131
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sched.step()
    assert abs(opt.lr - 0.001) < 1e-12
    sd = opt.state_dict()
    opt2 = ClipSGD(ps, 0.5, 1.0)
    opt2.load_state_dict(sd)
    assert abs(opt2.lr - 0.001) < 1e-12 and opt2.max_norm == 5.0
    ps[0].grad = torch.ones(5)
    opt.zero_grad()
    assert ps[0].grad is None
    with pytest.raises(Exception):
        ps[0].grad = torch.ones(5)
        ps[1].grad = torch.ones(3, 3)
        opt.step()                          # CPU tensors: no CPU path
