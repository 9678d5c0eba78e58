"""Host logic: config loader semantics, model factory surface, state_dict contract, weight packing."""
import os

import numpy as np
import pytest
import torch

from rdpn6d_amd.config import Config, gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer, fold_bn, pack_conv_weight, BNP


def test_config_base_merge_and_delete(tmp_path):
    (tmp_path / "base.py").write_text("A = dict(x=1, y=dict(p=1, q=2))\nS = dict(OPT=dict(type='RMSprop', lr=1, momentum=0))\n")
    (tmp_path / "child.py").write_text(
        "_base_ = './base.py'\nA = dict(y=dict(q=3), z=5)\nS = dict(OPT=dict(_delete_=True, type='Ranger', lr=1e-4))\n")
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.A.x == 1 and cfg.A.y.p == 1 and cfg.A.y.q == 3 and cfg.A.z == 5
    assert dict(cfg.S.OPT) == {"type": "Ranger", "lr": 1e-4}
    cfg.merge_from_dict(["A.y.q=7", "A.name=abc", "NEW.K=[1,2]"])
    assert cfg.A.y.q == 7 and cfg.A.name == "abc" and cfg.NEW.K == [1, 2]
    assert cfg.A.get("nope", 4) == 4


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference configs only exist in the build container")
def test_loads_reference_config_files_unchanged():
    cfg = Config.fromfile("/root/reference/configs/gdrn/lm/a6_cPnP_lm13.py")
    assert cfg.MODEL.CDPN.ROT_HEAD.NUM_REGIONS == 32 and cfg.MODEL.CDPN.PNP_NET.ROT_TYPE == "allo_rot6d"
    assert cfg.MODEL.CDPN.BACKBONE.NUM_LAYERS == 34 and cfg.SOLVER.OPTIMIZER_CFG.type == "Ranger"
    assert "momentum" not in cfg.SOLVER.OPTIMIZER_CFG  # _delete_=True honoured
    assert cfg.INPUT.FORMAT == "BGR" and cfg.TEST.USE_PNP is False
    ours = gdrn_base_cfg()
    for sec in ("BACKBONE", "ROT_HEAD", "PNP_NET"):
        for k, v in ours.MODEL.CDPN[sec].items():
            if k == "PRETRAINED":
                continue
            assert cfg.MODEL.CDPN[sec][k] == v, (sec, k)


def test_factory_surface():
    cfg = gdrn_base_cfg(device="cpu")
    model, opt = build_model_optimizer(cfg)
    assert "type" not in cfg.MODEL.CDPN.PNP_NET.PNP_HEAD_CFG  # popped in place like the reference
    assert len(model.state_dict()) == 305
    assert sum(p.numel() for p in model.parameters()) == 36403630
    sd = model.state_dict()
    assert tuple(sd["rot_head_net.features.0.weight"].shape) == (1024, 256, 3, 3)
    assert tuple(sd["rot_head_net.features.21.weight"].shape) == (37, 256, 1, 1)
    assert tuple(sd["pnp_net.features.0.weight"].shape) == (128, 43, 3, 3)
    assert tuple(sd["pnp_net.fc1.weight"].shape) == (1024, 8192)
    assert len(opt.param_groups) == 3
    bad = gdrn_base_cfg(device="cpu")
    bad.MODEL.CDPN.PNP_NET.ROT_TYPE = "nonsense"
    with pytest.raises(ValueError):
        build_model_optimizer(bad)
    bad = gdrn_base_cfg(device="cpu")
    bad.MODEL.CDPN.PNP_NET.PNP_HEAD_CFG.type = "Other"
    with pytest.raises(ValueError):
        build_model_optimizer(bad)
    # K = 64 (the reference cannot build this: nIn is hard-coded to 43)
    model64, _ = build_model_optimizer(gdrn_base_cfg(num_regions=64, device="cpu"))
    assert tuple(model64.state_dict()["pnp_net.features.0.weight"].shape) == (128, 75, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 6, 256, 256), do_loss=True, roi_coord_2d=torch.zeros(1, 5, 64, 64), fps=torch.zeros(1, 32, 3),
              roi_cams=torch.eye(3)[None])


def test_weight_packing_and_bn_fold():
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = pack_conv_weight(w)
    assert tuple(p.shape) == (64, 9, 16)
    assert p[1, 4, 2] == w[1, 2, 1, 1] and p[0, 0, 1] == w[0, 1, 0, 0]
    assert p[2:].abs().sum() == 0 and p[:, :, 3:].abs().sum() == 0
    bn = BNP(4)
    with torch.no_grad():
        bn.weight.copy_(torch.tensor([1.0, 2.0, 0.5, 1.5]))
        bn.bias.copy_(torch.tensor([0.1, -0.2, 0.3, 0.0]))
        bn.running_mean.copy_(torch.tensor([0.5, -1.0, 2.0, 0.0]))
        bn.running_var.copy_(torch.tensor([1.0, 4.0, 0.25, 9.0]))
    sc, sh = fold_bn(bn, conv_bias=torch.tensor([1.0, 1.0, 1.0, 1.0]))
    x = torch.randn(5, 4)
    ref = torch.nn.functional.batch_norm(x + 1.0, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, 1e-5)
    assert torch.allclose(x * sc[:4] + sh[:4], ref, atol=1e-6)
    assert sc.numel() == 64 and (sc[4:] == 1).all() and (sh[4:] == 0).all()


@pytest.mark.parametrize("layers,nparams_trunk_m", [(18, 11.2), (50, 23.5), (101, 42.5)])
def test_other_resnet_trunks_match_the_oracle_key_for_key(layers, nparams_trunk_m):
    """resnet_backbone.py:15-21 (BasicBlock 18 / 34, Bottleneck 50 / 101 / 152): same state_dict keys and shapes as the
    oracle's torchvision-style blocks (so torchvision://resnetNN checkpoints load), point-wise fusion input = layer4 width."""
    from oracle import model_oracle

    cfg = gdrn_base_cfg(device="cpu")
    cfg.MODEL.CDPN.BACKBONE.NUM_LAYERS = layers
    model, _ = build_model_optimizer(cfg)
    orc = model_oracle.GDRNOracle(32, "none", num_layers=layers)
    a, b = model.state_dict(), orc.state_dict()
    assert set(a) == set(b) and all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
    exp = 1 if layers < 50 else 4
    assert tuple(a["backbone.spatial_net.xyz_emb.weight"].shape) == (64, 512 * exp, 1, 1)
    trunk = sum(v.numel() for k, v in model.backbone.named_parameters() if k.startswith(("conv1", "bn1", "layer")))
    assert abs(trunk / 1e6 - nparams_trunk_m) < 0.1  # torchvision resnet18 / 50 / 101 without the fc layer
    bad = gdrn_base_cfg(device="cpu")
    bad.MODEL.CDPN.BACKBONE.NUM_LAYERS = 35
    with pytest.raises(ValueError):
        build_model_optimizer(bad)


@pytest.mark.parametrize("path,value", [
    ("MODEL.CDPN.ROT_HEAD.ROT_CLASS_AWARE", True), ("MODEL.CDPN.ROT_HEAD.REGION_CLASS_AWARE", True),
    ("MODEL.CDPN.ROT_HEAD.ROT_CONCAT", True), ("MODEL.CDPN.ROT_HEAD.NORM", "GN"), ("MODEL.CDPN.BACKBONE.INPUT_CHANNEL", 4),
    ("MODEL.CDPN.PNP_NET.TRANS_WITH_BOX_INFO", "ltrb"), ("MODEL.CDPN.USE_MTL", True), ("MODEL.CDPN.TRANS_HEAD.ENABLED", True),
])
def test_unsupported_switches_fail_loudly(path, value):
    """switches the reference accepts but the HIP path does not implement must not build a silently different network"""
    cfg = gdrn_base_cfg(device="cpu")
    node = cfg
    keys = path.split(".")
    for k in keys[:-1]:
        node = node[k]
    node[keys[-1]] = value
    with pytest.raises(NotImplementedError):
        build_model_optimizer(cfg)


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs/gdrn"), reason="needs the reference checkout (build container only)")
def test_every_shipped_rgbd_config_builds_through_the_factory(monkeypatch):
    """all model configs under configs/gdrn (138: K = 32 / 64, MASK_ATTENTION mul / none, 30 with SOLVER.AMP) load unchanged and
    build: same 305 state_dict keys, Ranger with three parameter groups.  BASE_LR is derived from OPTIMIZER_CFG the way
    main_gdrn.py:63-74 does before it calls the factory."""
    import glob

    # the shipped configs name BACKBONE.PRETRAINED = "torchvision://resnet34", which cannot be resolved offline: the factory then
    # raises unless told that a random trunk is acceptable (test_pretrained_backbone_loading covers the loading itself)
    monkeypatch.setenv("RDPN6D_ALLOW_RANDOM_BACKBONE", "1")
    n = amp = k64 = 0
    for f in sorted(glob.glob("/root/reference/configs/gdrn/**/*.py", recursive=True)):
        try:
            cfg = Config.fromfile(f)
        except ModuleNotFoundError:
            continue  # tlessSO/icp.py imports mmcv at module level
        if "MODEL" not in cfg:
            continue  # empty placeholder files
        cfg.MODEL.DEVICE = "cpu"
        model, opt = build_model_optimizer(cfg)
        assert len(model.state_dict()) == 305 and type(opt).__name__ == "Ranger" and len(opt.param_groups) == 3, f
        assert float(cfg.SOLVER.BASE_LR) == float(cfg.SOLVER.OPTIMIZER_CFG["lr"])
        n += 1
        amp += bool(cfg.SOLVER.AMP.ENABLED)
        k64 += int(cfg.MODEL.CDPN.ROT_HEAD.NUM_REGIONS) == 64
    assert n == 138 and amp == 30 and k64 == 10


def test_pretrained_backbone_loading(tmp_path, monkeypatch):
    """BACKBONE.PRETRAINED (GDRN.py:836-851 -> mmcv load_checkpoint(model.backbone, spec, strict=False)): a local .pth path and a
    torchvision:// name found in the torch hub cache load non-strictly into the trunk (fc ignored, fusion branch untouched);
    a spec that cannot be resolved offline RAISES instead of training from a random trunk unnoticed; MODEL.WEIGHTS != "" skips it."""
    import torch

    from rdpn6d_amd.gdrn import BackboneP

    donor = BackboneP(34)
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in donor.state_dict().items() if not k.startswith("spatial_net")}
    sd["fc.weight"], sd["fc.bias"] = torch.randn(1000, 512), torch.randn(1000)  # what an ImageNet file carries besides
    path = tmp_path / "resnet34-abc123.pth"
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    cfg = gdrn_base_cfg(device="cpu")
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED = str(path)
    model, _ = build_model_optimizer(cfg)
    got = model.backbone.state_dict()
    assert all(torch.equal(got[k], v) for k, v in sd.items() if k in got) and "fc.weight" not in got
    hub = tmp_path / "hub" / "checkpoints"
    hub.mkdir(parents=True)
    torch.save(sd, hub / "resnet34-b627a593.pth")
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    cfg = gdrn_base_cfg(device="cpu")
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED = "torchvision://resnet34"
    model, _ = build_model_optimizer(cfg)
    assert torch.equal(model.backbone.state_dict()["layer3.2.conv1.weight"], sd["layer3.2.conv1.weight"])
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nowhere"))
    cfg = gdrn_base_cfg(device="cpu")
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED = "torchvision://resnet34"
    with pytest.raises(FileNotFoundError):
        build_model_optimizer(cfg)
    cfg = gdrn_base_cfg(device="cpu")
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED, cfg.MODEL.WEIGHTS = "torchvision://resnet34", "output/some_checkpoint.pth"
    build_model_optimizer(cfg)  # a full checkpoint follows: the trunk initialisation is skipped like in the reference


def test_lr_schedule_none_and_step_when_annealing_starts_at_the_end():
    """anneal_point = 1 (anneal_start == total_iters) must not divide by zero for the methods that never use the fraction"""
    import torch

    from rdpn6d_amd.lr_scheduler import flat_and_anneal_lr_scheduler

    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    for method, kw in (("none", dict(anneal_point=1.0)), ("step", dict(steps=(1.0,)))):
        f = flat_and_anneal_lr_scheduler(opt, 100, anneal_method=method, **kw).lr_lambdas[0]
        assert f(100) in (1, 1.0, 0.1) and f(50) == 1


def test_image_sizes_of_the_2d3d_pnp_and_per_model_weight_epochs():
    """host logic of round 3 that needs no GPU: (a) the per-crop [H, W] table of the 2D-3D PnP (gdrn_evaluator.py:346-347: the batch's
    im_H / im_W) from scalars, lists or tensors, cfg.TEST.IM_H / IM_W only when BOTH are set, an error otherwise - never a silent
    480 x 640; (b) raw-pointer writers bump the epoch of the models that own the tensors they wrote, not every model's."""
    import torch

    from rdpn6d_amd import gdrn
    from rdpn6d_amd.config import ConfigDict, gdrn_base_cfg
    from rdpn6d_amd.gdrn import GDRN, build_model_optimizer

    dev = torch.device("cpu")
    hw = GDRN._image_sizes(480, 640, 3, dev, ConfigDict())
    assert hw.dtype == torch.int32 and hw.tolist() == [[480, 640]] * 3
    hw = GDRN._image_sizes(torch.tensor([480.0, 540.0]), [640, 720], 2, dev, ConfigDict())
    assert hw.tolist() == [[480, 640], [540, 720]] and hw.is_contiguous()
    assert GDRN._image_sizes(None, None, 2, dev, ConfigDict(IM_H=540, IM_W=720)).tolist() == [[540, 720]] * 2
    for bad in (dict(), dict(IM_H=480)):
        with pytest.raises(ValueError, match="im_H"):
            GDRN._image_sizes(None, None, 2, dev, ConfigDict(bad))
    with pytest.raises(ValueError):
        GDRN._image_sizes(0, 640, 2, dev, ConfigDict())

    a, _ = build_model_optimizer(gdrn_base_cfg(device="cpu"))
    b, _ = build_model_optimizer(gdrn_base_cfg(device="cpu"))
    sa, sb = a._weights_stamp(), b._weights_stamp()
    gdrn.bump_weights_epoch(list(a.pnp_net.parameters())[:2])  # what Ranger.step reports: the parameters it wrote
    assert a._weights_stamp() != sa and b._weights_stamp() == sb
    gdrn.bump_weights_epoch(b)                                  # what the train engine reports: its model
    assert b._weights_stamp() != sb
    sa = a._weights_stamp()
    with torch.no_grad():
        next(a.parameters()).add_(1.0)                          # torch in-place ops are seen through the version counters
    assert a._weights_stamp() != sa
    sa, sb = a._weights_stamp(), b._weights_stamp()
    gdrn.bump_weights_epoch()                                   # unknown writer: every live model
    assert a._weights_stamp() != sa and b._weights_stamp() != sb


def test_replaced_parameter_objects_and_model_copies_are_seen_by_the_plan_stamp():
    """(1) assigning a NEW Parameter / module object anywhere in the holder tree (invisible to the `_version` counters of the
    cached tensor list) changes the stamp GDRN.plan() compares; (2) a deepcopy / unpickled copy (EMA / teacher model) registers itself
    for the raw-pointer writers' epoch bumps and starts without the original's plans."""
    import copy
    import pickle

    import torch

    from rdpn6d_amd import gdrn
    from rdpn6d_amd.config import gdrn_base_cfg

    m, _ = gdrn.build_model_optimizer(gdrn_base_cfg(device="cpu"))
    s0 = m._weights_stamp()
    assert m._weights_stamp() == s0
    m.backbone.conv1.weight = torch.nn.Parameter(torch.zeros_like(m.backbone.conv1.weight))
    s1 = m._weights_stamp()
    assert s1 != s0 and id(m.backbone.conv1.weight) in m._tensor_ids()
    m.rot_head_net.features[3] = gdrn.ConvP(256, 256, 3, 1, 1)  # through the ModuleList
    s2 = m._weights_stamp()
    assert s2 != s1 and id(m.rot_head_net.features[3].weight) in m._tensor_ids()
    m.backbone.layer2[0].downsample[1].register_buffer("running_mean", torch.zeros(128))
    assert m._weights_stamp() != s2
    m._plans[("sentinel",)] = object()
    m2 = copy.deepcopy(m)
    assert m2 in gdrn._MODELS and m2._plans == {} and m2._h2_flags == {} and len(m2.state_dict()) == 305
    gdrn.bump_weights_epoch(list(m2.parameters())[:1])
    assert m2._weights_epoch == m._weights_epoch + 1
    m._plans.clear()
    m3 = pickle.loads(pickle.dumps(m))
    assert m3 in gdrn._MODELS and len(m3.state_dict()) == 305


def test_convpnpnet_h2_static_range_proof():
    """cfg.TEST.PNP_H2: ConvPnPNet's intermediate activations are stored in the h2 format only when the weights PROVE they fit
    (GroupNorm output bound, Cauchy-Schwarz over GroupNorm groups for fc1, spectral norm for fc2) - holds for He-scaled weights,
    fails when a GroupNorm gain is blown up."""
    import numpy as np
    import torch

    from rdpn6d_amd import gdrn, synth
    from rdpn6d_amd.config import gdrn_base_cfg

    m, _ = gdrn.build_model_optimizer(gdrn_base_cfg(device="cpu"))
    for mk in (synth.make_trained_like_state_dict, synth.make_state_dict):
        sd = mk({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        assert gdrn.InferencePlan._pnp_h2_range_ok(m.pnp_net, 64)
    with torch.no_grad():
        m.pnp_net.features[7].weight.mul_(100.0)
    assert not gdrn.InferencePlan._pnp_h2_range_ok(m.pnp_net, 64)
    with torch.no_grad():
        m.pnp_net.features[7].weight.div_(100.0)
        m.pnp_net.fc2.weight.mul_(50.0)
    assert not gdrn.InferencePlan._pnp_h2_range_ok(m.pnp_net, 64)


def test_sigma_max_bound_is_an_upper_bound_without_a_device_eigen_solver():
    """the spectral norm behind the ConvPnPNet range proof no longer goes through rocSOLVER on the device - a
    trace-power bound on the host (matrix products only), never below the true value (LAPACK on the CPU as the checker), within
    n^(1/128) of it, cached by weight content."""
    import torch

    from rdpn6d_amd import gdrn

    g = torch.Generator().manual_seed(3)
    for shape, scale in (((64, 512), 1.0), ((256, 96), 0.05), ((32, 32), 7.0)):
        w = torch.randn(*shape, generator=g) * scale
        true = float(torch.linalg.svdvals(w.double())[0])
        ub = gdrn._sigma_max_upper_bound(w)
        assert true <= ub <= true * min(shape) ** (1.0 / 256.0) * (1 + 1e-6), (shape, true, ub)
        assert gdrn._sigma_max_upper_bound(w.clone()) == ub and len(gdrn._SIGMA_MAX_CACHE) >= 1  # same content: cached
    low_rank = torch.outer(torch.arange(1.0, 9.0), torch.ones(40))  # one non-zero singular value: the bound is exact up to n^(1/k)
    assert abs(gdrn._sigma_max_upper_bound(low_rank) / float(torch.linalg.svdvals(low_rank.double())[0]) - 1.0) < 0.02
    assert gdrn._sigma_max_upper_bound(torch.zeros(4, 8)) == 0.0


def test_bench_gpus_flag_is_checked_before_anything_touches_a_gpu():
    """`bench.py --gpus N`: N must equal the launcher's WORLD_SIZE - checked before the first GPU call, so the
    rule is testable on a CPU box; a bare --gpus 1 on a GPU-less box still fails loudly (no CPU fallback)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "at least one rank" in r.stderr
    if not torch.cuda.is_available():
        env.pop("WORLD_SIZE")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_fps_python_face_returns_what_the_reference_returns(oracle_lib, monkeypatch):
    """`ops.farthest_point_sampling` is the body of core/csrc/fps/fps_utils.py:6-21: it returns pts[idxs] ALONE, so that the body of
    get_fps_and_center (core/utils/data_utils.py:217-226: concatenate with the centroid row) runs on it unchanged.  On this GPU-less
    box the C ABI behind the face is stood in for by the C oracle (the checker; the product library needs a GPU and is exercised by
    tests/test_gpu_kernels.py::test_get_fps_and_center_body_on_the_hip_face)."""
    import ctypes
    from types import SimpleNamespace

    import numpy as np

    from rdpn6d_amd import _lib, ops
    from tests.fps_cases import make_cloud

    def fps_host(pts, idx, pn, sn, start):
        if start < 0:
            oracle_lib.oracle_fps_init_center(pts, idx, pn, sn)
        else:
            oracle_lib.oracle_fps_from_start(pts, idx, pn, sn, start)
        return 0

    monkeypatch.setattr(_lib, "load", lambda: SimpleNamespace(rdpn6d_fps_host=fps_host, rdpn6d_last_error=lambda: b""))
    pts = make_cloud("gauss", 3000, 21).astype(np.float64)  # the loaders hand float64 model points over
    num_fps = 8
    # -- body of get_fps_and_center, with the package's function where the reference imports its own
    avgx, avgy, avgz = np.average(pts[:, 0]), np.average(pts[:, 1]), np.average(pts[:, 2])
    fps_pts = ops.farthest_point_sampling(pts, num_fps, init_center=True)
    res_pts = np.concatenate([fps_pts, np.array([[avgx, avgy, avgz]])], axis=0)
    # --
    assert isinstance(fps_pts, np.ndarray) and fps_pts.shape == (num_fps, 3) and fps_pts.dtype == np.float32
    assert res_pts.shape == (num_fps + 1, 3)
    want = np.zeros(num_fps, np.int32)
    p32 = np.ascontiguousarray(pts, np.float32)
    oracle_lib.oracle_fps_init_center(p32.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), 3000, num_fps)
    assert np.array_equal(fps_pts, p32[want])
    both = ops.farthest_point_sampling(pts, num_fps, init_center=True, return_index=True)
    assert isinstance(both, tuple) and np.array_equal(both[1], want) and np.array_equal(both[0], fps_pts)


def test_h2_overflow_lowers_only_the_flagged_tensors(monkeypatch):
    """GDRN._lower_h2_exponents (round 6): the range-flag array has one slot per launch; the slots that were raised name exponent
    variables through the plan's table - those go down two binades, nothing else moves; a raised slot WITHOUT a variable (glue row,
    depth-xyz), a plan that is not the all-h2 one, or a variable at the floor make it decline (the caller then leaves h2 as before).
    Host logic only: the flag tensors are stood in for by CPU tensors and a recorded event."""
    from types import SimpleNamespace

    import torch

    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    model, _ = build_model_optimizer(gdrn_base_cfg(device="cpu"))
    dev = torch.device("cpu")
    key = model._dev_key(dev)
    ev = SimpleNamespace(synchronize=lambda: None, query=lambda: True)

    def arm(slots):
        host = torch.zeros(model.NFLAG, dtype=torch.int32)
        host[list(slots)] = 1
        model._h2_flags[key] = [torch.zeros(model.NFLAG, dtype=torch.int32), host, ev]

    plan = SimpleNamespace(_slots=["stem", "layer2", None, "rot_head.convT", "layer2"], h2_pointwise=True)
    arm([1, 4])  # two launches of one residual chain
    assert model._lower_h2_exponents(plan, dev) and model.h2_exponents(dev) == {"layer2": 2}
    assert int(model._h2_flags[key][1].abs().sum()) == 0 and model._h2_flags[key][2] is None  # flags cleared for the re-run
    arm([1, 3])
    assert model._lower_h2_exponents(plan, dev) and model.h2_exponents(dev) == {"layer2": 0, "rot_head.convT": 2}
    arm([2])  # a launch without a variable
    before = dict(model.h2_exponents(dev))
    assert not model._lower_h2_exponents(plan, dev) and model.h2_exponents(dev) == before
    arm([0, 2])  # ... even next to one with a variable: nothing is lowered
    assert not model._lower_h2_exponents(plan, dev) and model.h2_exponents(dev) == before
    arm([400])  # past the plan's slots (the shared last slot)
    assert not model._lower_h2_exponents(plan, dev)
    arm([0])
    assert not model._lower_h2_exponents(SimpleNamespace(_slots=plan._slots, h2_pointwise=False), dev)
    model.h2_exponents(dev)["stem"] = -12  # at the floor
    assert not model._lower_h2_exponents(plan, dev) and model.h2_exponents(dev)["stem"] == -12
    arm([])  # nothing raised
    assert not model._lower_h2_exponents(plan, dev)
    # load_state_dict forgets a calibration made for other weights; a deep copy keeps it
    import copy

    twin = copy.deepcopy(model)
    assert twin.h2_exponents(dev) == model.h2_exponents(dev) and twin.h2_exponents(dev) is not model.h2_exponents(dev)
    model.load_state_dict(model.state_dict())
    assert model.h2_exponents(dev) == {}
