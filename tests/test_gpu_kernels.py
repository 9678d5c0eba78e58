"""GPU parity tests (run on the MI355X box: ``pytest -m gpu``).  Every kernel is called through the
C ABI of librdpn6d_hip.so and compared with a plain PyTorch-CPU fp32 reference of the same op
(floating-point kernels) or with the C oracle / golden vectors (fps: bit-exact indices)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from rdpn6d_amd import _lib

    lib = _lib.load()
    assert lib.rdpn6d_device_count() >= 1
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(ref, 1.0), f"{what}: max abs err {err:.3e} (ref max {ref:.3e})"


CONV_CASES = [
    # B, H, Cin, Cout, k, stride, pad, res, act
    (2, 16, 16, 64, 3, 1, 1, False, 1),
    (2, 16, 64, 64, 3, 1, 1, True, 1),
    (3, 16, 64, 128, 3, 2, 1, False, 1),
    (3, 16, 64, 128, 1, 2, 0, False, 0),
    (2, 8, 256, 256, 3, 1, 1, True, 1),
    (1, 64, 256, 256, 3, 1, 1, False, 1),   # large M: 128x128 tiles
    (2, 32, 256, 37, 1, 1, 0, False, 0),    # ragged N
    (2, 64, 48, 128, 3, 2, 1, False, 0),
    (5, 1, 512, 1024, 1, 1, 0, False, 2),   # FC-like, leaky
    (2, 7, 32, 64, 3, 1, 1, False, 1),      # odd spatial, M not multiple of the tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm(dev, case):
    from rdpn6d_amd import ops

    B, H, Cin, Cout, k, stride, pad, use_res, act = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w, None, stride, pad) * scale[None, :, None, None] + shift[None, :, None, None]
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.1)
    y = ops.conv2d_nhwc(nhwc(x).to(dev), w.to(dev), scale.to(dev), shift.to(dev), stride, pad,
                        residual=nhwc(res).to(dev) if use_res else None, act=act, slope=0.1)
    torch.cuda.synchronize()
    _close(nchw(y), ref, 2e-5, f"conv {case}")


def test_conv_channel_slices(dev):
    """input read from a channel slice, output written into a slice of a wider buffer (free concat)."""
    from rdpn6d_amd import ops

    g = torch.Generator().manual_seed(5)
    xfull = torch.randn(2, 10, 10, 48, generator=g)
    w = torch.randn(20, 16, 1, 1, generator=g)
    out = torch.full((2, 10, 10, 40), 7.0)
    o = ops.conv2d_nhwc(xfull.to(dev), w.to(dev), in_co=16, out=out.to(dev), out_co=8)
    torch.cuda.synchronize()
    ref = F.conv2d(nchw(xfull[..., 16:32]), w)
    _close(nchw(o[..., 8:28]), ref, 2e-5, "slice conv")
    assert (o[..., :8] == 7).all() and (o[..., 28:] == 7).all()


def test_force_tiles(dev):
    """every tile configuration gives the same answer."""
    from rdpn6d_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 12, 12, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    ref = F.conv2d(x, w, None, 1, 1)
    try:
        for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
            lib.rdpn6d_conv_force_tile(bm, bn)
            y = ops.conv2d_nhwc(nhwc(x).to(dev), w.to(dev), stride=1, pad=1)
            torch.cuda.synchronize()
            _close(nchw(y), ref, 2e-5, f"tile {bm}x{bn}")
    finally:
        lib.rdpn6d_conv_force_tile(0, 0)


def test_stem(dev):
    from rdpn6d_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 64, 64, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x[:, :3], w, None, 2, 3) * sc[None, :, None, None] + sh[None, :, None, None])
    y = ops.stem_conv7x7(x.to(dev), w.to(dev), sc.to(dev), sh.to(dev))
    torch.cuda.synchronize()
    _close(nchw(y), ref, 2e-5, "stem")


def test_maxpool_upsample_gmax_gn(dev):
    from rdpn6d_amd import ops

    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 18, 18, generator=g)
    _close(nchw(ops.maxpool3x3s2(nhwc(x).to(dev))), F.max_pool2d(x, 3, 2, 1), 0, "maxpool")
    x = torch.randn(2, 32, 8, 8, generator=g)
    _close(nchw(ops.upsample_bilinear(nhwc(x).to(dev), 4)), F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=True),
           2e-6, "upsample")
    x = torch.randn(3, 64, 6, 5, generator=g)
    buf = torch.zeros(3, 6, 5, 128)
    buf[..., :64] = nhwc(x)
    o = ops.global_max_concat_(buf.to(dev), 64).cpu()
    assert torch.equal(o[..., 64:], x.amax(dim=(2, 3))[:, None, None, :].expand(3, 6, 5, 64))
    assert torch.equal(o[..., :64], nhwc(x))
    x = torch.randn(3, 128, 8, 8, generator=g) * 2 + 0.3
    ga, be = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = F.relu(F.group_norm(x, 32, ga, be, 1e-5))
    y = ops.groupnorm_relu_(nhwc(x).to(dev), 32, ga.to(dev), be.to(dev))
    _close(nchw(y), ref, 1e-5, "groupnorm")


def test_fps_bit_exact(dev, oracle_lib, golden_dir):
    from rdpn6d_amd import ops
    from tests.fps_cases import fps_cases, make_cloud

    gold = np.load(os.path.join(golden_dir, "fps_golden.npz"))
    for name, kind, n, sn, seed, mode in fps_cases():
        pts = make_cloud(kind, n, seed)
        if mode == "center":
            _, idx = ops.farthest_point_sampling(pts, sn, init_center=True, return_index=True)
        else:
            _, idx = ops.farthest_point_sampling(pts, sn, start=int(mode), return_index=True)
        assert np.array_equal(idx, gold[name]), f"{name}: {idx[:8]} vs {gold[name][:8]}"


def test_fps_many_workgroups_per_cloud_bit_identical(dev):
    """clouds above 16 384 points: rdpn6d_fps_device_ws spreads a cloud over ceil(N / 16 384) workgroups (points in registers, one
    cross-workgroup barrier per sample); the indices are those of the single-workgroup kernel bit for bit - a batch of three clouds of
    different sizes (one of them a lattice full of exact distance ties), bounding-box-centre and fixed starts; no barrier timed out."""
    from rdpn6d_amd import _lib
    from tests.fps_cases import make_cloud

    lib = _lib.load()
    clouds = [make_cloud("gauss", 50000, 3), make_cloud("lattice", 70001, 4), make_cloud("sphere", 20000, 5)]
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32)
    pts = torch.from_numpy(np.concatenate(clouds)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    sn, nobj, max_pn = 48, 3, int(max(len(c) for c in clouds))
    md = torch.empty(int(off[-1]), device=dev)
    ws = torch.zeros(int(lib.rdpn6d_fps_workspace_bytes(nobj)), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    for start in (-1, 0, 12345):
        one = torch.full((nobj, sn), -7, dtype=torch.int32, device=dev)
        many = torch.full((nobj, sn), -9, dtype=torch.int32, device=dev)
        _lib.check(lib.rdpn6d_fps_device(P(pts), P(d_off), nobj, max_pn, sn, start, P(one), P(md), st))
        _lib.check(lib.rdpn6d_fps_device_ws(P(pts), P(d_off), nobj, max_pn, sn, start, P(many), P(md), P(ws), ws.numel(), st))
        torch.cuda.synchronize()
        assert int(ws.view(torch.int32).view(nobj, -1)[:, 1].abs().sum()) == 0, "a cross-workgroup barrier timed out"
        assert torch.equal(one, many), (start, one[:, :6], many[:, :6])
        assert int(one.min()) >= 0


def test_fps_host_abi_survives_a_barrier_timeout(dev, oracle_lib, monkeypatch):
    """the reference-named symbols send clouds above 16 384 points to the many-workgroup kernel, whose spin barrier assumes
    the cloud's workgroups are resident together.  When it gives up (simulated: RDPN6D_FPS_TEST_TIMEOUT sets the error word and the
    -1 indices it leaves) the host path re-runs the cloud on one workgroup instead of failing - same indices as the C oracle."""
    from rdpn6d_amd import _lib
    from tests.fps_cases import make_cloud

    lib = _lib.load()
    P = ctypes.c_void_p
    pts = make_cloud("gauss", 40000, 8)
    want = np.zeros(24, dtype=np.int32)
    oracle_lib.oracle_fps_init_center(pts.ctypes.data_as(P), want.ctypes.data_as(P), 40000, 24)
    monkeypatch.setenv("RDPN6D_FPS_TEST_TIMEOUT", "1")
    idx = np.full(24, -5, dtype=np.int32)
    lib.farthest_point_sampling_init_center(pts.ctypes.data_as(P), idx.ctypes.data_as(P), 40000, 24)
    assert np.array_equal(idx, want), (idx[:8], want[:8])


def test_fps_reference_symbols(dev, oracle_lib):
    """the two reference-named void symbols (ext.h) with host pointers."""
    from rdpn6d_amd import _lib
    from tests.fps_cases import make_cloud

    lib = _lib.load()
    P = ctypes.c_void_p
    pts = make_cloud("gauss", 3000, 21)
    idx = np.zeros(16, dtype=np.int32)
    lib.farthest_point_sampling_init_center(pts.ctypes.data_as(P), idx.ctypes.data_as(P), 3000, 16)
    want = np.zeros(16, dtype=np.int32)
    oracle_lib.oracle_fps_init_center(pts.ctypes.data_as(P), want.ctypes.data_as(P), 3000, 16)
    assert np.array_equal(idx, want)
    lib.farthest_point_sampling(pts.ctypes.data_as(P), idx.ctypes.data_as(P), 3000, 16)  # random start
    oracle_lib.oracle_fps_from_start(pts.ctypes.data_as(P), want.ctypes.data_as(P), 3000, 16, int(idx[0]))
    assert np.array_equal(idx, want)


def test_get_fps_and_center_body_on_the_hip_face(dev, oracle_lib):
    """the body of get_fps_and_center (core/utils/data_utils.py:217-226) on ops.farthest_point_sampling: the face returns pts[idxs]
    alone, like core/csrc/fps/fps_utils.py:21."""
    from rdpn6d_amd import ops
    from tests.fps_cases import make_cloud

    pts = make_cloud("gauss", 3000, 21).astype(np.float64)
    fps_pts = ops.farthest_point_sampling(pts, 8, init_center=True)
    res_pts = np.concatenate([fps_pts, np.array([[np.average(pts[:, 0]), np.average(pts[:, 1]), np.average(pts[:, 2])]])], axis=0)
    p32 = np.ascontiguousarray(pts, np.float32)
    want = np.zeros(8, dtype=np.int32)
    oracle_lib.oracle_fps_init_center(p32.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), 3000, 8)
    assert res_pts.shape == (9, 3) and np.array_equal(res_pts[:8].astype(np.float32), p32[want])


# ----------------------------------------------------------------------------- whole path vs golden
@pytest.fixture(scope="module")
def golden_setup(dev, golden_dir):
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    gold = np.load(os.path.join(golden_dir, "model_c1.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1.npz"))
    inp = synth.make_inputs(4, seed=0)
    assert synth.sha256_of([inp[k] for k in sorted(inp)]) == str(gold["sha256_inputs"])
    models = {}
    for att in ("none", "mul"):
        cfg = gdrn_base_cfg(mask_attention=att, device="cuda")
        model, _ = build_model_optimizer(cfg)
        sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
        sd.update({k: bn[k] for k in bn.files})
        assert synth.sha256_of([sd[k] for k in sorted(sd) if not k.endswith("num_batches_tracked")]) == str(gold["sha256_weights"])
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        model.eval()
        models[att] = model
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    return models, t, gold


def _run(model, t):
    with torch.no_grad():
        o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"],
                  roi_centers=t["roi_center"], roi_whs=t["roi_wh"], roi_extents=t["roi_extent"],
                  resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"], im_H=480, im_W=640)  # (synth.IM_H, IM_W: the 2D-3D PnP scales coord2d by them)
    torch.cuda.synchronize()
    return o


@pytest.fixture(scope="module")
def truth(golden_setup):
    """fp64 evaluation of the (reference-pinned) oracle on the C1 batch = the exact answer.

    Why: this 45-layer random-weight network amplifies fp32 round-off ~100x from stem to head.  The
    REFERENCE's own fp32 output (the golden file) sits 6.5e-4 (max-abs, maps) / 2-3e-4 (pose, rel) away
    from the exact answer, so no independent fp32 implementation - including the reference on another
    CPU - can be closer to the golden file than that.  Tolerances below are therefore stated as
    2x the reference's own fp32 error, measured here against fp64, not as a bare 1e-4."""
    from oracle import model_oracle

    models, t, gold = golden_setup
    out = {}
    for att in ("none", "mul"):
        o64 = model_oracle.GDRNOracle(32, att)
        o64.load_state_dict({k: v.cpu() for k, v in models[att].state_dict().items()}, strict=True)
        o64.double().eval()
        tc = {k: (v.cpu().double() if v.dtype.is_floating_point else v.cpu()) for k, v in t.items()}
        with torch.no_grad():
            out[att] = o64(tc["roi_img"], tc["roi_coord_2d"], tc["fps"], tc["roi_cam"], tc["roi_center"], tc["roi_wh"],
                           tc["resize_ratio"])
    return out


def test_model_maps_vs_reference_golden(golden_setup, truth):
    """tier (i): dense maps vs the reference's own outputs (fp32 path)."""
    models, t, gold = golden_setup
    o = _run(models["none"], t)
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        mine, ref, exact = o[k].cpu().numpy().astype(np.float64), gold["eval_" + k].astype(np.float64), truth["none"][k].numpy()
        ref_self = np.abs(ref - exact).max()
        e_gold, e_exact = np.abs(mine - ref).max(), np.abs(mine - exact).max()
        relf = np.linalg.norm(mine - ref) / np.linalg.norm(ref)
        print(f"{k}: HIP-vs-reference {e_gold:.3e} (rel-Frobenius {relf:.2e}) | HIP-vs-fp64 {e_exact:.3e} | reference-vs-fp64 {ref_self:.3e}")
        ref_relf = np.linalg.norm(ref - exact) / np.linalg.norm(exact)
        assert e_gold <= 2.0 * ref_self and e_exact <= 2.0 * ref_self, k
        assert relf <= 2.0 * ref_relf, (k, relf, ref_relf)
    am = models["none"].plan(4, t["roi_img"].device).argmax.cpu().numpy().reshape(4, 64, 64)
    agree = (am == gold["eval_region_argmax"]).mean()
    print("region arg-max agreement with the reference", agree)
    assert agree > 0.998


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("att", ["none", "mul"])
def test_model_pose_vs_reference_golden(golden_setup, truth, att):
    """tier (iii): end-to-end pose (batch Frobenius and worst sample) vs the reference."""
    models, t, gold = golden_setup
    o = _run(models[att], t)
    R, T = gold[f"eval_{att}_rot"].astype(np.float64), gold[f"eval_{att}_trans"].astype(np.float64)
    Rx, Tx = truth[att]["rot"].numpy().astype(np.float64), truth[att]["trans"].numpy()
    r, tr = o["rot"].cpu().numpy().astype(np.float64), o["trans"].cpu().numpy().astype(np.float64)
    ref_self_r = max(_rel(R[i], Rx[i]) for i in range(4))
    ref_self_t = max(_rel(T[i], Tx[i]) for i in range(4))
    worst_r = max(_rel(r[i], R[i]) for i in range(4))
    worst_t = max(_rel(tr[i], T[i]) for i in range(4))
    print(f"[{att}] pose rel err vs reference: R {_rel(r, R):.3e} (worst {worst_r:.3e})  t {_rel(tr, T):.3e} (worst {worst_t:.3e})"
          f" | reference-vs-fp64 worst: R {ref_self_r:.3e} t {ref_self_t:.3e}"
          f" | HIP-vs-fp64 worst: R {max(_rel(r[i], Rx[i]) for i in range(4)):.3e} t {max(_rel(tr[i], Tx[i]) for i in range(4)):.3e}")
    assert worst_r <= 2.0 * max(ref_self_r, 1e-4) and worst_t <= 2.0 * max(ref_self_t, 1e-4)
    assert np.allclose(np.linalg.det(r), 1.0, atol=1e-5)


def test_pose_decode_kernel_alone(dev):
    """teacher-forced tier (ii): the pose kernel on the reference's own (rot6d, t) head outputs is
    well inside the north star's 1e-4 (no network round-off in the way)."""
    import ctypes
    from rdpn6d_amd import _lib, synth
    from rdpn6d_amd.gdrn import _ptr

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_c1.npz"))
    inp = synth.make_inputs(4, seed=0)
    lib = _lib.load()
    for att in ("none", "mul"):
        rt = torch.zeros(4, 16)
        rt[:, :6] = torch.from_numpy(gold[f"eval_{att}_pred_rot6d"])
        rt[:, 6:9] = torch.from_numpy(gold[f"eval_{att}_pred_t_"])
        rt = rt.to(dev)
        rot, trans = torch.empty(4, 3, 3, device=dev), torch.empty(4, 3, device=dev)
        g = {k: torch.from_numpy(inp[k]).to(dev) for k in ("roi_cam", "roi_center", "roi_wh", "resize_ratio")}
        _lib.check(lib.rdpn6d_pose_decode_f32(_ptr(rt), 16, _ptr(g["roi_cam"]), _ptr(g["roi_center"]), _ptr(g["roi_wh"]),
                                              _ptr(g["resize_ratio"]), 4, 1, 0, _ptr(rot), _ptr(trans), None))
        torch.cuda.synchronize()
        er = _rel(rot.cpu().numpy(), gold[f"eval_{att}_rot"])
        et = _rel(trans.cpu().numpy(), gold[f"eval_{att}_trans"])
        print(f"[{att}] pose kernel on reference head outputs: R {er:.2e} t {et:.2e}")
        assert er < 2e-6 and et < 2e-6


def test_model_vs_oracle_other_batch(golden_setup, dev):
    """a batch the golden file does not hold (B=3, other seed): HIP vs the torch-CPU oracle, anchored
    on the fp64 evaluation like above."""
    from oracle import model_oracle
    from rdpn6d_amd import synth

    models, _, _ = golden_setup
    model = models["mul"]
    orc = model_oracle.GDRNOracle(32, "mul")
    orc.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
    orc.eval()
    inp = synth.make_inputs(3, seed=7)
    tc = {k: torch.from_numpy(v) for k, v in inp.items()}
    args = lambda d: (d["roi_img"], d["roi_coord_2d"], d["fps"], d["roi_cam"], d["roi_center"], d["roi_wh"], d["resize_ratio"])
    with torch.no_grad():
        oo = orc(*args(tc))
        o64 = orc.double()(*args({k: (v.double() if v.dtype.is_floating_point else v) for k, v in tc.items()}))
    o = _run(model, {k: v.to(dev) for k, v in tc.items()})
    for k in ("mask", "coor_x", "region"):
        self_err = (oo[k].double() - o64[k]).abs().max().item()
        err = (o[k].cpu().double() - oo[k].double()).abs().max().item()
        print(f"{k}: HIP-vs-oracle {err:.3e}, oracle fp32-vs-fp64 {self_err:.3e}")
        assert err <= 2.0 * self_err
    # discrete arg-max flips (a one-ulp logit difference moves a pixel to another anchor) are the one
    # place where fp32 round-off turns into an O(1) change of a ConvPnPNet input, so pose parity is
    # stated per sample together with the number of flipped pixels
    am_hip = model.plan(3, dev).argmax.cpu().numpy().reshape(3, -1)
    am32, am64 = oo["region_argmax"].numpy().reshape(3, -1), o64["region_argmax"].numpy().reshape(3, -1)
    for i in range(3):
        fl_hip, fl_ref = int((am_hip[i] != am32[i]).sum()), int((am32[i] != am64[i]).sum())
        er = _rel(o["rot"][i].cpu().numpy().astype(np.float64), oo["rot"][i].numpy().astype(np.float64))
        et = _rel(o["trans"][i].cpu().numpy().astype(np.float64), oo["trans"][i].numpy().astype(np.float64))
        sr = _rel(oo["rot"][i].numpy().astype(np.float64), o64["rot"][i].numpy())
        st_ = _rel(oo["trans"][i].numpy().astype(np.float64), o64["trans"][i].numpy())
        print(f"sample {i}: arg-max flips HIP-vs-oracle {fl_hip}, oracle fp32-vs-fp64 {fl_ref} | pose rel err HIP-vs-oracle R {er:.2e} t {et:.2e}"
              f" | oracle fp32-vs-fp64 R {sr:.2e} t {st_:.2e}")
        assert fl_hip <= 8
        tol_r, tol_t = (2.0 * max(sr, 1e-4), 2.0 * max(st_, 1e-4)) if fl_hip == 0 and fl_ref == 0 else (1e-2, 1e-2)
        assert er <= tol_r and et <= tol_t, i


# ----------------------------------------------------------------------------- RANSAC / Kabsch
@pytest.mark.parametrize("outliers,K,side", [(0.0, 32, 64), (0.3, 32, 64), (0.6, 32, 64), (0.4, 64, 64), (0.3, 8, 80)])
def test_ransac_bit_exact_vs_oracle_and_ground_truth(dev, oracle_lib, outliers, K, side):
    """inlier masks, counts and the winning hypothesis are bit-exact vs the C oracle under a fixed seed;
    the refit pose agrees to 1e-5; both recover the known pose."""
    from rdpn6d_amd import ops
    from tests.ransac_cases import make_case, pose_errors
    from tests.test_ransac_oracle import run_oracle

    c = make_case(B=5, K=K, side=side, outliers=outliers, seed=17 + K + side)
    for seed in (1, 99):
        pose_o, nin_o, msk_o, best_o = run_oracle(oracle_lib, c, seed=seed)
        g = {k: torch.from_numpy(c[k]).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
        pose, nin, msk, best = ops.ransac_kabsch(g["out_nchw"], g["coord2d"], g["fps"], g["extents"], g["ratios"], g["argmax"],
                                                 seed=seed)
        torch.cuda.synchronize()
        assert np.array_equal(best.cpu().numpy(), best_o), (best.cpu().numpy(), best_o)
        assert np.array_equal(nin.cpu().numpy(), nin_o)
        assert np.array_equal(msk.cpu().numpy(), msk_o)
        assert np.abs(pose.cpu().numpy() - pose_o).max() < 1e-5
        for b in range(c["B"]):
            re, te = pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])
            assert re < 0.5 and te < 0.002, (b, re, te)


def test_ransac_sentinel_and_edge_cases(dev, oracle_lib):
    from rdpn6d_amd import ops
    from tests.ransac_cases import make_case
    from tests.test_ransac_oracle import run_oracle

    c = make_case(B=3, seed=3)
    c["out_nchw"][:2, 0] = 0.0
    c["out_nchw"][:2, 0, 0], c["out_nchw"][:2, 0, 1] = -1.0, 1.0
    c["out_nchw"][1, 0, 5] = 0.9           # crop 0: no point, crop 1: one point, crop 2: normal
    c["out_nchw"][2, 0, :] = np.where(np.arange(4096) % 7 == 0, c["out_nchw"][2, 0, :], 0.0)
    c["out_nchw"][2, 0, 0], c["out_nchw"][2, 0, 1] = -0.2, 1.2
    po, ni, mo, bo = run_oracle(oracle_lib, c)
    g = {k: torch.from_numpy(c[k]).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
    pose, nin, msk, best = ops.ransac_kabsch(g["out_nchw"], g["coord2d"], g["fps"], g["extents"], g["ratios"], g["argmax"], seed=7)
    torch.cuda.synchronize()
    assert (pose[:2].cpu().numpy() == -100).all() and (nin[:2].cpu().numpy() == 0).all()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(msk.cpu().numpy(), mo)
    assert np.abs(pose.cpu().numpy() - po).max() < 1e-5


def test_model_use_pnp_matches_oracle_on_model_maps(golden_setup, dev, oracle_lib):
    """TEST.USE_PNP=True: the RANSAC/Kabsch launched inside GDRN.forward on the model's own maps equals
    the C oracle run on those same maps (bit-exact masks / counts), and returns the extra out_dict keys."""
    from tests.test_ransac_oracle import run_oracle

    models, t, _ = golden_setup
    model = models["none"]
    model.cfg.TEST.USE_PNP, model.cfg.TEST.PNP_TYPE = True, "ransac_kabsch"
    model.cfg.TEST.PNP_INLIER_THR = 0.05
    try:
        o = _run(model, t)
    finally:
        model.cfg.TEST.USE_PNP, model.cfg.TEST.PNP_TYPE = False, "ransac_pnp"
    plan = model.plan(4, dev)
    c = dict(out_nchw=plan.out_nchw.cpu().numpy().reshape(4, 37, 4096), coord2d=t["roi_coord_2d"].cpu().numpy().reshape(4, 5, 4096),
             fps=t["fps"].cpu().numpy(), extents=t["roi_extent"].cpu().numpy(), ratios=t["resize_ratio"].cpu().numpy(),
             argmax=plan.argmax.cpu().numpy(), B=4, HW=4096, K=32)
    po, ni, mo, bo = run_oracle(oracle_lib, c, inlier_thr=0.05, seed=0)
    assert np.array_equal(o["pnp_num_inliers"].cpu().numpy(), ni)
    assert np.array_equal(o["pnp_inlier_mask"].cpu().numpy(), mo)
    assert np.abs(o["pnp_pose"].cpu().numpy() - po).max() < 1e-4
    assert o["pnp_pose"].shape == (4, 12)


@pytest.mark.parametrize("att", ["none", "mul"])
def test_pose_teacher_forced_on_reference_maps(golden_setup, dev, att):
    """tier (ii) of SURVEY.md section 8d: feed the REFERENCE's dense maps (golden) into the HIP glue -> ConvPnPNet -> pose decode,
    so that trunk/head round-off and arg-max flips are out of the picture: the pose then agrees with the reference to well
    inside the north star's 1e-4 (measured ~1e-6), i.e. the remaining end-to-end gap is the 45-layer fp32 round-off, not the pose path."""
    import ctypes
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _ptr

    models, t, gold = golden_setup
    model = models[att]
    plan = model.plan(4, dev)
    _run(model, t)  # builds every buffer, leaves real maps behind - now overwrite the head output with the reference's
    maps = np.concatenate([gold["eval_mask"], gold["eval_coor_x"], gold["eval_coor_y"], gold["eval_coor_z"], gold["eval_region"]], 1)  # (4,37,64,64)
    ho = plan.bufs["head_out"]
    ho.zero_()
    ho[:, :, :37] = torch.from_numpy(maps).to(dev).reshape(4, 37, 4096).permute(0, 2, 1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(plan.glue_fn(*plan.glue_args(t["roi_coord_2d"].contiguous(), t["fps"].contiguous()), st), "glue")  # (h2 plan: writes ConvPnPNet's input as an h2 record)
    for L in plan.post:
        _lib.check(L.fn(*L.args, st), L.name)
    g = {k: t[k].float().contiguous() for k in ("roi_cam", "roi_center", "roi_wh", "resize_ratio")}
    _lib.check(plan.lib.rdpn6d_pose_decode_f32(_ptr(plan.rt), 16, _ptr(g["roi_cam"]), _ptr(g["roi_center"]), _ptr(g["roi_wh"]),
                                               _ptr(g["resize_ratio"]), 4, 1, 0, _ptr(plan.rot), _ptr(plan.trans), st), "pose")
    torch.cuda.synchronize()
    assert np.array_equal(plan.argmax.cpu().numpy().reshape(4, 64, 64), gold["eval_region_argmax"])  # same maps -> same anchors
    R, T = gold[f"eval_{att}_rot"].astype(np.float64), gold[f"eval_{att}_trans"].astype(np.float64)
    r, tr = plan.rot.cpu().numpy().astype(np.float64), plan.trans.cpu().numpy().astype(np.float64)
    wr = max(_rel(r[i], R[i]) for i in range(4))
    wt = max(_rel(tr[i], T[i]) for i in range(4))
    print(f"[{att}] teacher-forced pose rel err (worst sample): R {wr:.2e} t {wt:.2e}")
    assert wr < 1e-4 and wt < 1e-4
    assert np.abs(plan.rt[:, :6].cpu().numpy() - gold[f"eval_{att}_pred_rot6d"]).max() < 1e-4


@pytest.mark.parametrize("K,R,cam", [(64, 256, "lm"), (32, 320, "ycbv"), (8, 128, "lm")])
def test_generalised_geometry_vs_oracle(dev, few_threads, K, R, cam):
    """NUM_REGIONS != 32 and INPUT_RES != 256 (10 LM-O configs use K=64; C5 uses 320x320): the reference hard-codes nIn=43 and
    the 64x64 / 8x8 geometry and cannot build these; the HIP path is checked against the (generalised) torch-CPU oracle with the
    same fp64 yardstick as C1."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    cfg = gdrn_base_cfg(num_regions=K, mask_attention="mul", device="cuda", num_classes=21 if cam == "ycbv" else 13)
    cfg.MODEL.CDPN.BACKBONE.INPUT_RES, cfg.MODEL.CDPN.BACKBONE.OUTPUT_RES = R, R // 4
    model, _ = build_model_optimizer(cfg)
    orc = model_oracle.GDRNOracle(K, "mul", out_res=R // 4)
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in orc.state_dict().items()}, seed=4321)
    orc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    inp = synth.make_inputs(2, seed=11, res=R, num_regions=K, cam=cam)
    tc = {k: torch.from_numpy(v) for k, v in inp.items()}
    model_oracle.calibrate_bn(orc, tc["roi_img"])
    model.load_state_dict(orc.state_dict(), strict=True)
    model.eval()
    args = lambda d: (d["roi_img"], d["roi_coord_2d"], d["fps"], d["roi_cam"], d["roi_center"], d["roi_wh"], d["resize_ratio"])  # noqa: E731
    with torch.no_grad():
        o32 = orc(*args(tc))
        o64 = orc.double()(*args({k: (v.double() if v.dtype.is_floating_point else v) for k, v in tc.items()}))
    o = _run(model, {k: v.to(dev) for k, v in tc.items()})
    assert o["region"].shape == (2, K + 1, R // 4, R // 4) and o["mask"].shape == (2, 1, R // 4, R // 4)
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        self_err = (o32[k].double() - o64[k]).abs().max().item()
        err = (o[k].cpu().double() - o32[k].double()).abs().max().item()
        print(f"K={K} R={R} {k}: HIP-vs-oracle {err:.3e}, oracle fp32-vs-fp64 {self_err:.3e}")
        assert err <= 2.5 * self_err
    am = model.plan(2, dev).argmax.cpu().numpy().reshape(2, -1)
    flips = int((am != o32["region_argmax"].numpy().reshape(2, -1)).sum())
    er = _rel(o["rot"].cpu().numpy().astype(np.float64), o32["rot"].numpy().astype(np.float64))
    et = _rel(o["trans"].cpu().numpy().astype(np.float64), o32["trans"].numpy().astype(np.float64))
    print(f"K={K} R={R}: arg-max flips {flips}, pose rel err R {er:.2e} t {et:.2e}")
    assert flips <= 8 and er < (2e-3 if flips == 0 else 2e-2) and et < (2e-3 if flips == 0 else 2e-2)


# ----------------------------------------------------------------------------- bf16 mode (cfg.TEST.AMP_TEST)
BF16_CONV_CASES = [
    # B, H, Cin, Cout, k, stride, res, act
    (2, 16, 64, 64, 3, 1, True, 1),
    (3, 16, 64, 128, 3, 2, False, 1),
    (3, 16, 64, 128, 1, 2, False, 0),
    (1, 64, 256, 256, 3, 1, False, 1),    # 128x128 tiles, RB=128
    (2, 32, 96, 128, 1, 1, False, 1),     # Cin % 64 != 0 -> 64-byte K-chunks
    (2, 32, 256, 37, 1, 1, False, 0),     # ragged N, fp32 output only
    (2, 7, 32, 64, 3, 1, True, 1),        # odd spatial size, M not a multiple of the tile
    (5, 1, 512, 1024, 1, 1, False, 2),    # FC-like, leaky
]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_conv_igemm_bf16(dev, case):
    """bf16 matrix-pipe kernel vs the fp32 kernel on the SAME bf16-valued operands: products of bf16 numbers are exact
    in fp32, so the two differ by fp32 summation order only (1e-5); the bf16 store is one RNE rounding of that."""
    from rdpn6d_amd import ops

    B, H, Cin, Cout, k, stride, use_res, act = case
    g = torch.Generator().manual_seed(sum(case) * 7 + 1)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev).bfloat16()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(dev).bfloat16() if use_res else None
    kw = dict(stride=stride, pad=k // 2, act=act, slope=0.1)
    ref = ops.conv2d_nhwc(x.float(), w.float(), sc, sh, residual=res.float() if use_res else None, **kw)
    y32 = ops.conv2d_nhwc(x, w, sc, sh, residual=res.float() if use_res else None, out_f32=True, **kw)  # fp32 out <-> fp32 residual
    torch.cuda.synchronize()
    _close(y32, ref, 1e-5, "bf16 conv, fp32 store")
    if Cout % 8 == 0:
        y16 = ops.conv2d_nhwc(x, w, sc, sh, residual=res, **kw)
        assert y16.dtype == torch.bfloat16
        # one rounding of (almost) the same fp32 value: at most one bf16 ulp where the fp32 values straddle a tie
        d = (y16.float() - ref.bfloat16().float()).abs()
        assert (d <= ref.abs() * 2.0 ** -7 + 1e-5).all()  # (+1e-5: fp32 values that differ by summation order around 0)
        assert (d > 0).float().mean().item() < 1e-2


def test_bf16_pointwise_kernels(dev):
    import ctypes
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _ptr

    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 14, 14, 64, generator=g).to(dev).bfloat16()
    y = torch.empty(3, 7, 7, 64, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.rdpn6d_maxpool3x3s2_bf16(_ptr(x), 3, 14, 14, 64, _ptr(y), st))
    ref = F.max_pool2d(nchw(x.float()), 3, 2, 1)
    assert torch.equal(nchw(y.float()), ref)  # max commutes with rounding: exact
    u = torch.empty(3, 56, 56, 64, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.rdpn6d_upsample_bilinear_bf16(_ptr(x), 3, 14, 14, 64, 4, _ptr(u), st))
    ref = F.interpolate(nchw(x.float()), scale_factor=4, mode="bilinear", align_corners=True)
    assert (nchw(u.float()) - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() * 1.01
    buf = torch.zeros(2, 9, 9, 256, device=dev, dtype=torch.bfloat16)
    buf[..., :128] = torch.randn(2, 9, 9, 128, generator=g).to(dev).bfloat16()
    want = buf[..., :128].float().amax(dim=(1, 2))
    _lib.check(lib.rdpn6d_global_max_concat_bf16(_ptr(buf), 2, 81, 128, 256, st))
    assert torch.equal(buf[..., 128:].float(), want[:, None, None, :].expand(2, 9, 9, 128))
    img = torch.rand(2, 6, 32, 32, generator=g).to(dev)
    xs = torch.zeros(2, 4, 4, 96, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.rdpn6d_xyz_subsample_bf16(_ptr(img), 2, 6, 32, 8, _ptr(xs), 96, 64, st))
    assert torch.equal(xs[..., 64:67].float(), img[:, 3:6, ::8, ::8].permute(0, 2, 3, 1).bfloat16().float())
    assert xs[..., :64].abs().max().item() == 0 and xs[..., 67:].abs().max().item() == 0
    w = torch.randn(64, 3, 7, 7, generator=g).to(dev) * 0.1
    sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    from rdpn6d_amd import ops
    y32 = ops.stem_conv7x7(img, w, sc, sh)
    y16 = torch.empty(2, 16, 16, 64, device=dev, dtype=torch.bfloat16)
    wp = w.permute(0, 2, 3, 1).contiguous()
    _lib.check(lib.rdpn6d_stem_conv7x7_bf16(_ptr(img), 2, 6, 32, _ptr(wp), _ptr(sc), _ptr(sh), _ptr(y16), st))
    torch.cuda.synchronize()
    assert torch.equal(y16, y32.bfloat16())


def test_bf16_mode_vs_autocast_yardstick(dev):
    """cfg.TEST.AMP_TEST=True (the reference's autocast switch, gdrn_evaluator.py:625): trunk + fusion + head on the bf16
    matrix pipe.  Yardstick = the torch-CPU oracle under torch.autocast(bfloat16) - what the reference's own autocast
    path computes with an 8-bit mantissa - both measured against the fp64 evaluation.  The seeded random-weight network
    amplifies round-off ~2000x (fp32 already loses 4 digits), which turns ANY 8-bit-mantissa run into noise, so this test
    uses residual branches damped x0.1 (bn2.weight), where the amplification is ~100x.  Assertion: the HIP bf16 maps are
    at least as close to the exact answer as the autocast oracle's (the HIP path keeps the head output, the glue,
    ConvPnPNet and the pose decode in fp32)."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    orc = model_oracle.GDRNOracle(32, "none")
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in orc.state_dict().items()}, seed=1234)
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd.items()}
    for k in sd:
        if k.endswith("bn2.weight"):
            sd[k] *= 0.1
    orc.load_state_dict(sd, strict=True)
    inp = synth.make_inputs(4, seed=0)
    tc = {k: torch.from_numpy(v) for k, v in inp.items()}
    model_oracle.calibrate_bn(orc, tc["roi_img"])
    orc.eval()
    cfg = gdrn_base_cfg(mask_attention="none", device="cuda")
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(orc.state_dict(), strict=True)
    model.eval()
    args = lambda d: (d["roi_img"], d["roi_coord_2d"], d["fps"], d["roi_cam"], d["roi_center"], d["roi_wh"], d["resize_ratio"])  # noqa: E731
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.bfloat16):
            oac = orc(*args(tc))
        o64 = orc.double()(*args({k: (v.double() if v.dtype.is_floating_point else v) for k, v in tc.items()}))
    t = {k: v.to(dev) for k, v in tc.items()}
    o32 = _run(model, t)
    o32 = {k: o32[k].clone() for k in ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans")}
    model.cfg.TEST.AMP_TEST = True
    try:
        o16 = _run(model, t)
        assert model.plan(4, dev).bf16 and model.plan(4, dev).bufs["head_a"].dtype == torch.bfloat16
    finally:
        model.cfg.TEST.AMP_TEST = False
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        exact = o64[k]
        sc = exact.abs().max().item()
        e16 = (o16[k].cpu().double() - exact).abs().max().item() / sc
        eac = (oac[k].double() - exact).abs().max().item() / sc
        e32 = (o32[k].cpu().double() - exact).abs().max().item() / sc
        f16 = (torch.linalg.norm(o16[k].cpu().double() - exact) / torch.linalg.norm(exact)).item()
        fac = (torch.linalg.norm(oac[k].double() - exact) / torch.linalg.norm(exact)).item()
        print(f"{k}: max-abs/scale HIP-bf16 {e16:.3e} | autocast oracle {eac:.3e} | HIP-fp32 {e32:.3e} ; rel-Frobenius HIP-bf16 {f16:.3e} | autocast {fac:.3e}")
        assert e16 <= 1.1 * eac and f16 <= 1.1 * fac, k
    for k in ("rot", "trans"):
        e16, eac = _rel(o16[k].cpu().numpy().astype(np.float64), o64[k].numpy()), _rel(oac[k].float().numpy().astype(np.float64), o64[k].numpy())
        print(f"{k}: rel err HIP-bf16 {e16:.3e} | autocast oracle {eac:.3e}")


@pytest.mark.parametrize("M,K,N,ks", [(64, 8192, 1024, 32), (32, 1024, 256, 8), (5, 512, 9, 3), (64, 8192, 1024, 1000)])
def test_conv_splitk_matches_fused(dev, M, K, N, ks):
    """split-K FC path (partial sums + fixed-order reduce + epilogue) vs the single-pass kernel: same products, different
    fp32 summation order."""
    import ctypes
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _pad_to, _pad_vec, _ptr

    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.zeros(_pad_to(N, 64), 1, K, device=dev)
    w[:N, 0] = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    sh = _pad_vec(torch.randn(N, generator=g).to(dev), w.shape[0], 0.0)
    res = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for split in (1, ks):
        y = torch.full((M, _pad_to(N, 4)), 7.0, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.shift, d.res, d.y = _ptr(x), _ptr(w), _ptr(sh), _ptr(res), _ptr(y)
        d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride, d.ntaps = M, 1, 1, K, K, 1, 1, 1, 1
        d.N, d.Npad, d.OH, d.OW, d.osy, d.osx = N, w.shape[0], 1, 1, 1, 1
        d.out_cs, d.res_cs, d.act, d.slope = y.shape[1], N, 2, 0.1
        ws = torch.empty(max(1, int(lib.rdpn6d_conv_splitk_ws_floats(ctypes.byref(d), split))), device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.rdpn6d_conv2d_splitk_f32(ctypes.byref(d), split, _ptr(ws), st))
        torch.cuda.synchronize()
        outs.append(y)
    ref = F.leaky_relu(x.double() @ w[:N, 0].double().t() + sh[:N].double() + res.double(), 0.1)
    _close(outs[0][:, :N], ref, 2e-6, "fused")
    _close(outs[1][:, :N], ref, 2e-6, "split-K")
    assert (outs[1][:, N:] == 7.0).all()  # padding columns untouched


def test_hip_graph_replay_is_bit_identical(golden_setup, dev):
    """cfg.TEST.HIP_GRAPH: first call eager, second call captures, third replays - same kernels in the same order, so every
    output is bit-identical to the eager step; new input buffers fall back to eager / a new capture."""
    models, t, _ = golden_setup
    model = models["mul"]
    ref = _run(model, t)
    ref = {k: v.clone() for k, v in ref.items() if v is not None}
    model.cfg.TEST.HIP_GRAPH = True
    model.cfg.TEST.USE_PNP = True
    model.cfg.TEST.PNP_INLIER_THR = 0.05
    try:
        eager = {k: v.clone() for k, v in _run(model, t).items() if v is not None}
        outs = [_run(model, t) for _ in range(3)]
        plan = model.plan(4, dev)
        assert any(isinstance(g, torch.cuda.CUDAGraph) for g in plan._graphs.values())
        for o in outs:
            for k, v in eager.items():
                assert torch.equal(o[k], v), k
        for k in ("mask", "coor_x", "region", "rot", "trans"):
            assert torch.equal(eager[k], ref[k]), k
        t2 = {k: v.clone() for k, v in t.items()}
        t2["roi_img"] = t2["roi_img"].flip(0).contiguous()
        o2 = _run(model, t2)  # other buffers: eager again, and the old graph must not have been replayed on them
        assert torch.equal(o2["mask"], ref["mask"].flip(0))
    finally:
        model.cfg.TEST.HIP_GRAPH = False
        model.cfg.TEST.USE_PNP = False


@pytest.mark.parametrize("mode,iters", [("ransac", 20), ("ransac", 100), ("iter", 1)])
def test_ransac_net_initialised_bit_exact_vs_oracle(dev, oracle_lib, mode, iters):
    """A10 (process_net_and_pnp): network pose as hypothesis 0 / all-points fit, fallbacks and the translation guard -
    masks, counts and winner bit-exact vs the C oracle, poses to 1e-5."""
    from rdpn6d_amd import ops
    from tests.ransac_cases import make_case
    from tests.test_ransac_oracle import _gt_pose12, run_oracle_net

    c = make_case(B=6, outliers=0.35 if mode == "ransac" else 0.0, seed=31)
    net = _gt_pose12(c)
    net[1, :9] = np.eye(3).reshape(9)            # crop 1: wrong rotation
    net[2, 9:] += np.float32(1.5)                # crop 2: translation guard fires
    c["out_nchw"][3, 0] = 0.0                    # crop 3: no correspondences -> network pose
    c["out_nchw"][3, 0, 0], c["out_nchw"][3, 0, 1] = -1.0, 1.0
    m = {"ransac": 1, "iter": 2}[mode]
    po, ni, mo, bo = run_oracle_net(oracle_lib, c, net, mode=m, iters=iters, seed=5)
    g = {k: torch.from_numpy(c[k]).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
    pose, nin, msk, best = ops.ransac_kabsch(g["out_nchw"], g["coord2d"], g["fps"], g["extents"], g["ratios"], g["argmax"],
                                             iters=iters, seed=5, net_pose=torch.from_numpy(net).to(dev), net_mode=mode)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(nin.cpu().numpy(), ni)
    assert np.array_equal(msk.cpu().numpy(), mo)
    assert np.abs(pose.cpu().numpy() - po).max() < 1e-5
    assert np.array_equal(pose[3].cpu().numpy(), net[3]) and np.array_equal(pose[2, 9:].cpu().numpy(), net[2, 9:])


@pytest.mark.parametrize("pnp_type", ["net_ransac_kabsch", "net_iter_kabsch"])
def test_model_pnp_type_net_variants(golden_setup, dev, oracle_lib, pnp_type):
    """cfg.TEST.PNP_TYPE (gdrn_evaluator.py:136-145) inside GDRN.forward: equals the C oracle run on the model's own maps and
    its own decoded pose."""
    from tests.test_ransac_oracle import run_oracle_net

    models, t, _ = golden_setup
    model = models["none"]
    model.cfg.TEST.USE_PNP, model.cfg.TEST.PNP_TYPE, model.cfg.TEST.PNP_INLIER_THR = True, pnp_type, 0.05
    try:
        o = _run(model, t)
    finally:
        model.cfg.TEST.USE_PNP, model.cfg.TEST.PNP_TYPE = False, "ransac_pnp"
    plan = model.plan(4, dev)
    c = dict(out_nchw=plan.out_nchw.cpu().numpy().reshape(4, 37, 4096), coord2d=t["roi_coord_2d"].cpu().numpy().reshape(4, 5, 4096),
             fps=t["fps"].cpu().numpy(), extents=t["roi_extent"].cpu().numpy(), ratios=t["resize_ratio"].cpu().numpy(),
             argmax=plan.argmax.cpu().numpy(), B=4, HW=4096, K=32)
    net = np.concatenate([o["rot"].cpu().numpy().reshape(4, 9), o["trans"].cpu().numpy()], 1)
    po, ni, mo, bo = run_oracle_net(oracle_lib, c, net, mode=1 if pnp_type == "net_ransac_kabsch" else 2, inlier_thr=0.05, iters=20, seed=0)
    assert np.array_equal(o["pnp_num_inliers"].cpu().numpy(), ni) and np.array_equal(o["pnp_inlier_mask"].cpu().numpy(), mo)
    assert np.abs(o["pnp_pose"].cpu().numpy() - po).max() < 1e-4


@pytest.mark.parametrize("layers,R,B", [(50, 256, 2), (50, 320, 2), (18, 256, 2), (18, 256, 16), (50, 256, 16)])
def test_other_resnet_trunks_vs_oracle(dev, few_threads, layers, R, B):
    """resnet_backbone.py:15-21 offers 18 / 34 (BasicBlock) and 50 / 101 / 152 (Bottleneck).  The reference cannot RUN the
    Bottleneck trunks (md_pointnet(512, ...) is hard-coded while layer4 then has 2048 channels) - BASELINE config 5 asks for
    ResNet-50 at 320x320 - so parity is against the generalised torch-CPU oracle, with the usual fp64 yardstick.
    B = 16 puts the trunk (BasicBlock or Bottleneck) and the head on the bf16x3 kernels (fp32 accuracy on the bf16 matrix pipe)."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    cfg.MODEL.CDPN.BACKBONE.NUM_LAYERS = layers
    cfg.MODEL.CDPN.BACKBONE.INPUT_RES, cfg.MODEL.CDPN.BACKBONE.OUTPUT_RES = R, R // 4
    model, _ = build_model_optimizer(cfg)
    orc = model_oracle.GDRNOracle(32, "mul", out_res=R // 4, num_layers=layers)
    assert set(orc.state_dict()) == set(model.state_dict())
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in orc.state_dict().items()}, seed=77)
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd.items()}
    for k in sd:  # the last BatchNorm of each residual branch damped, as a trained network has it (Goyal et al. zero-gamma
        if k.endswith("bn3.weight") or (layers < 50 and k.endswith("bn2.weight")):  # init): keeps round-off growth moderate
            sd[k] *= 0.25
    orc.load_state_dict(sd, strict=True)
    inp = synth.make_inputs(B, seed=5, res=R)
    tc = {k: torch.from_numpy(v) for k, v in inp.items()}
    model_oracle.calibrate_bn(orc, tc["roi_img"])
    model.load_state_dict(orc.state_dict(), strict=True)
    model.eval()
    args = lambda d: (d["roi_img"], d["roi_coord_2d"], d["fps"], d["roi_cam"], d["roi_center"], d["roi_wh"], d["resize_ratio"])  # noqa: E731
    with torch.no_grad():
        o32 = orc(*args(tc))
        o64 = orc.double()(*args({k: (v.double() if v.dtype.is_floating_point else v) for k, v in tc.items()}))
    o = _run(model, {k: v.to(dev) for k, v in tc.items()})
    plan = model.plan(B, dev)
    assert plan.x3_trunk and plan.x3_launches > 0  # h2: the trunk leaves the fp32 MFMA from one crop on
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        self_err = (o32[k].double() - o64[k]).abs().max().item()
        err = (o[k].cpu().double() - o64[k]).abs().max().item()
        print(f"resnet{layers} R={R} B={B} {k}: HIP-vs-fp64 {err:.3e}, oracle fp32-vs-fp64 {self_err:.3e}")
        assert err <= 2.5 * self_err + 1e-6
    am = plan.argmax.cpu().numpy().reshape(B, -1)
    flips = int((am != o64["region_argmax"].numpy().reshape(B, -1)).sum())
    er = _rel(o["rot"].cpu().numpy().astype(np.float64), o64["rot"].numpy())
    et = _rel(o["trans"].cpu().numpy().astype(np.float64), o64["trans"].numpy())
    print(f"resnet{layers} R={R} B={B}: arg-max flips vs fp64 {flips}, pose rel err R {er:.2e} t {et:.2e}")
    assert flips <= 4 * B and er < (2e-3 if flips == 0 else 2e-2) and et < (2e-3 if flips == 0 else 2e-2)
    if B == 2 and R == 256 and plan.fast == "h2" and plan.h2_pointwise:
        # (round 6) the per-tensor h2 exponents on THIS trunk's plan - Bottleneck: conv1 / conv2 variables, the residual chain through
        # conv3 and the down-sampling branch: every variable one binade down is the same network up to round-off
        names = sorted({v for v in plan._slots if v is not None})
        assert any(v.endswith(".conv1") for v in names) and {"layer2", "layer3", "layer4"} <= set(names)
        assert (layers < 50) or any(v.endswith(".conv2") for v in names)
        model.h2_exponents(dev).update({v: 3 for v in names})
        o3 = _run(model, {k: v.to(dev) for k, v in tc.items()})
        assert model.plan(B, dev) is not plan and not model.h2_range_exceeded(dev)
        for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
            self_err = (o32[k].double() - o64[k]).abs().max().item()
            err = (o3[k].cpu().double() - o64[k]).abs().max().item()
            assert err <= 3.0 * self_err + 1e-6, (k, err, self_err)


@pytest.mark.parametrize("case", [(2, 8, 512, 512, 3, True, 6), (1, 16, 256, 256, 3, False, 4), (3, 8, 96, 128, 1, False, 2)])
def test_conv_bf16_splitk_matches_fused(dev, case):
    """split-K form of the bf16 kernel (fp32 partials + fixed-order reduce + epilogue) vs the single-pass kernel: bf16 and fp32
    outputs agree to fp32 summation order (one bf16 ulp on rounding ties)."""
    import ctypes
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _pad_to, _pad_vec, _ptr, pack_conv_weight

    lib = _lib.load()
    B, H, Cin, Cout, k, use_res, ks = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, H, H, Cin, generator=g).to(dev).bfloat16()
    w = pack_conv_weight((torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev), cin_pad=_pad_to(Cin, 32)).bfloat16()
    sc = _pad_vec((torch.rand(Cout, generator=g) + 0.5).to(dev), w.shape[0], 1.0)
    sh = _pad_vec(torch.randn(Cout, generator=g).to(dev), w.shape[0], 0.0)
    res16 = torch.randn(B, H, H, Cout, generator=g).to(dev).bfloat16() if use_res else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = k // 2
    for out_f32 in (0, 1):
        outs = []
        res = (res16.float() if out_f32 else res16) if use_res else None
        for split in (1, ks):
            y = torch.zeros(B, H, H, Cout, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
            d = _lib.ConvDesc()
            d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(x), _ptr(w), _ptr(sc), _ptr(sh), _ptr(res), _ptr(y)
            d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, Cin, Cin, H, H, 1
            taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
            d.ntaps = len(taps)
            for t, (dy, dx) in enumerate(taps):
                d.dy[t], d.dx[t] = dy, dx
            d.N, d.Npad, d.OH, d.OW, d.osy, d.osx = Cout, w.shape[0], H, H, 1, 1
            d.out_cs, d.res_cs, d.act = Cout, Cout, 1
            ws = torch.empty(max(1, int(lib.rdpn6d_conv_splitk_ws_floats(ctypes.byref(d), split))), device=dev)
            _lib.check(lib.rdpn6d_conv2d_splitk_bf16(ctypes.byref(d), out_f32, split, _ptr(ws), st))
            torch.cuda.synchronize()
            outs.append(y.float())
        tol = 1e-5 if out_f32 else 2.0 ** -7
        assert ((outs[0] - outs[1]).abs() <= tol * outs[0].abs() + 1e-5).all(), (case, out_f32)
        assert outs[0].abs().max().item() > 0.1


BF16_8PH_CASES = [
    # B, H, Cin, Cout, k, stride, res
    (4, 64, 256, 256, 3, 1, True),     # the dense head's layer shape (64 full tiles)
    (3, 30, 128, 256, 3, 1, False),    # M = 2700: last tile ragged, odd spatial size
    (2, 32, 128, 512, 1, 1, True),     # 1x1, two K-tiles only (prologue + tail), two column tiles
    (2, 32, 128, 256, 3, 2, False),    # stride 2
    (1, 16, 1024, 256, 3, 1, False),   # long K (144 K-tiles), single tile
]


BF16_PP_CASES = [
    # B, H, Cin, Cout, k, stride, res
    (64, 16, 256, 256, 3, 1, True),    # layer3 at B = 64: 256 tiles of 128x128, four stages
    (64, 32, 128, 128, 3, 1, True),    # layer2 at B = 64: 256 tiles of 256x128, three stages
    (37, 32, 128, 128, 3, 1, False),   # 296 tiles of 128x128: two rounds, ragged over the XCDs
    (31, 30, 128, 256, 3, 1, True),    # M = 27 900: ragged last row tile, odd spatial size
    (64, 32, 512, 128, 1, 1, False),   # 1x1: 8 chunks, the shortest loop the kernel takes
    (64, 32, 128, 256, 3, 2, False),   # stride 2
]


@pytest.mark.parametrize("case", BF16_PP_CASES)
def test_conv_bf16_pingpong_kernel_bit_identical_to_128_tile(dev, case):
    """the eight-wave ping-pong kernel (csrc/conv_igemm_bf16_pp.hip) accumulates every output in the same order as the 128x128 tile kernel
    (K-chunks in order, four k16 steps each): the same bits as the forced tile - fp32 and 16-bit stores, with / without residual - on
    every one of six launches (race screen for the counted-vmcnt / two-group barrier schedule)."""
    import ctypes
    from rdpn6d_amd import _lib, ops

    lib = _lib.load()
    B, H, Cin, Cout, k, stride, use_res = case
    g = torch.Generator().manual_seed(sum(case) * 5 + 3)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev).bfloat16()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(dev).bfloat16() if use_res else None
    kw = dict(stride=stride, pad=k // 2, act=1, slope=0.1)
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride, d.ntaps, d.N, d.Npad, d.out_cs, d.res_cs = B, H, H, Cin, Cin, Ho, Ho, stride, k * k, Cout, Cout, Cout, Cout
    assert lib.rdpn6d_conv_bf16_uses_pingpong(ctypes.byref(d), 0) == 1 and lib.rdpn6d_conv_bf16_uses_pingpong(ctypes.byref(d), 1) == 1
    try:
        for out_f32 in (True, False):
            r = (res.float() if out_f32 else res) if use_res else None
            lib.rdpn6d_conv_bf16_force_tile(128, 128)
            want = ops.conv2d_nhwc(x, w, sc, sh, residual=r, out_f32=out_f32, **kw)
            torch.cuda.synchronize()
            lib.rdpn6d_conv_bf16_force_tile(0, 0)
            for rep in range(6):
                got = ops.conv2d_nhwc(x, w, sc, sh, residual=r, out_f32=out_f32, **kw)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (case, out_f32, rep, (got.float() - want.float()).abs().max().item())
            assert want.float().abs().max().item() > 0.1
    finally:
        lib.rdpn6d_conv_bf16_force_tile(0, 0)


@pytest.mark.parametrize("case", BF16_8PH_CASES)
def test_conv_bf16_8phase_kernel_bit_identical_to_128_tile(dev, case):
    """the 256x256 8-phase ping-pong kernel accumulates every output in the same order as the 128x128 kernel (K-tiles in
    order, four k16 steps each), so forcing either tile must give the SAME bits - fp32 and bf16 stores, with / without
    residual - and repeated launches must reproduce them (race screen for the counted-vmcnt / barrier schedule)."""
    from rdpn6d_amd import _lib, ops

    lib = _lib.load()
    B, H, Cin, Cout, k, stride, use_res = case
    g = torch.Generator().manual_seed(sum(case) * 3 + 5)
    x = torch.randn(B, H, H, Cin, generator=g).to(dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev).bfloat16()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).to(dev).bfloat16() if use_res else None
    kw = dict(stride=stride, pad=k // 2, act=1, slope=0.1)
    try:
        for out_f32 in (True, False):
            r = (res.float() if out_f32 else res) if use_res else None
            lib.rdpn6d_conv_bf16_force_tile(128, 128)
            want = ops.conv2d_nhwc(x, w, sc, sh, residual=r, out_f32=out_f32, **kw)
            torch.cuda.synchronize()
            lib.rdpn6d_conv_bf16_force_tile(256, 256)
            for rep in range(6):
                got = ops.conv2d_nhwc(x, w, sc, sh, residual=r, out_f32=out_f32, **kw)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (case, out_f32, rep, (got.float() - want.float()).abs().max().item())
            assert want.float().abs().max().item() > 0.1
    finally:
        lib.rdpn6d_conv_bf16_force_tile(0, 0)


def test_bf16_model_with_8phase_head_is_bit_identical(dev):
    """whole bf16 inference step at B=16 (the head layers, incl. the ConvTranspose phases, then pick the 8-phase kernel)
    vs the same step with the 128x128 tile forced: identical maps and poses."""
    from rdpn6d_amd import _lib, synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    lib = _lib.load()
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    cfg.TEST.AMP_TEST = True
    model, _ = build_model_optimizer(cfg)
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    bn = np.load(os.path.join(os.path.dirname(__file__), "golden", "bn_stats_c1.npz"))
    sd.update({k: bn[k] for k in bn.files})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(16, seed=3).items()}
    outs = []
    try:
        for tile in ((128, 128), (0, 0)):
            lib.rdpn6d_conv_bf16_force_tile(*tile)
            bm, bn = ctypes_tile_of_head(lib, 16)
            o = _run(model, inp)
            torch.cuda.synchronize()
            outs.append(({k: o[k].clone() for k in ("mask", "coor_x", "region", "rot", "trans")}, (bm, bn)))
    finally:
        lib.rdpn6d_conv_bf16_force_tile(0, 0)
    assert outs[0][1] == (128, 128) and outs[1][1] == (256, 256), (outs[0][1], outs[1][1])
    for k in outs[0][0]:
        assert torch.isfinite(outs[0][0][k]).all(), k
        assert torch.equal(outs[0][0][k].cpu(), outs[1][0][k].cpu()), k


def ctypes_tile_of_head(lib, B):
    """tile the bf16 dispatcher picks for the head's 3x3 / 256-channel layers at batch B"""
    import ctypes

    from rdpn6d_amd import _lib

    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride, d.ntaps, d.N, d.Npad = B, 64, 64, 256, 256, 64, 64, 1, 9, 256, 256
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.rdpn6d_conv_bf16_tile_for(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn)))
    return bm.value, bn.value


@pytest.mark.parametrize("bf16", [False, True])
def test_full_bench_batch_b64_properties(golden_setup, truth, dev, bf16):
    """BASELINE configuration C2 (B = 64 per GPU, what bench.py times) through size-independent properties: the batch is
    16 copies of the reference's four golden crops in a shuffled order, so
      * every copy of a crop must give the SAME bits (one crop = one independent unit; no cross-crop coupling, no
        position-dependent summation order) - maps, poses and, with TEST.USE_PNP, the RANSAC/Kabsch result;
      * rotations of the learned head are proper (orthonormal, det +1);
      * in fp32 the four golden crops must still meet the B=4 bounds against the reference's golden maps and poses (the
        kernels pick other tiles / split-K factors at this size)."""
    models, t, gold = golden_setup
    model = models["none"]
    rng = np.random.default_rng(5)
    order = np.concatenate([np.arange(4), rng.permutation(np.repeat(np.arange(4), 15))])  # crop id of each batch slot
    idx = torch.from_numpy(order).to(dev)
    t64 = {k: (v[idx].contiguous() if v.shape[0] == 4 else v) for k, v in t.items()}
    first = {c: int(np.where(order == c)[0][0]) for c in range(4)}
    tcfg = model.cfg.TEST
    old_amp, old_pnp = tcfg.get("AMP_TEST", False), tcfg.get("USE_PNP", False)

    def copies_identical(o, keys):
        for k in keys:
            v = o[k].cpu()
            assert torch.isfinite(v).all(), k
            for s in range(64):
                assert torch.equal(v[s], v[first[int(order[s])]]), (k, s, int(order[s]))

    try:
        tcfg.AMP_TEST, tcfg.USE_PNP = bf16, False
        o = _run(model, t64)
        o = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
        copies_identical(o, ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans"))
        R = o["rot"].cpu().double()
        eye = torch.eye(3, dtype=torch.float64).expand(64, 3, 3)
        assert (R @ R.transpose(1, 2) - eye).abs().max().item() < 1e-5 and (torch.linalg.det(R) - 1).abs().max().item() < 1e-5
        if not bf16:
            for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
                mine = o[k].cpu().numpy().astype(np.float64)[:4]
                ref, exact = gold["eval_" + k].astype(np.float64), truth["none"][k].numpy()
                ref_self = np.abs(ref - exact).max()
                assert np.abs(mine - ref).max() <= 2.0 * ref_self and np.abs(mine - exact).max() <= 2.0 * ref_self, k
            Rg, Tg = gold["eval_none_rot"].astype(np.float64), gold["eval_none_trans"].astype(np.float64)
            Rx, Tx = truth["none"]["rot"].numpy().astype(np.float64), truth["none"]["trans"].numpy()
            r, tr = o["rot"].cpu().numpy().astype(np.float64)[:4], o["trans"].cpu().numpy().astype(np.float64)[:4]
            worst_r, worst_t = max(_rel(r[i], Rg[i]) for i in range(4)), max(_rel(tr[i], Tg[i]) for i in range(4))
            ref_r, ref_t = max(_rel(Rg[i], Rx[i]) for i in range(4)), max(_rel(Tg[i], Tx[i]) for i in range(4))
            print(f"B=64: worst pose rel err of the golden crops vs the reference: R {worst_r:.2e} t {worst_t:.2e}")
            assert worst_r <= 2.0 * max(ref_r, 1e-4) and worst_t <= 2.0 * max(ref_t, 1e-4)
        tcfg.USE_PNP = True
        o = _run(model, t64)
        copies_identical({k: v.clone() for k, v in o.items() if torch.is_tensor(v)}, ("rot", "trans"))
    finally:
        tcfg.AMP_TEST, tcfg.USE_PNP = old_amp, old_pnp


BF16X3_CASES = [
    # B, H, Cin, Cout, k, stride, res, act
    (2, 32, 256, 256, 3, 1, True, 1),     # head layer shape (8 tiles)
    (3, 30, 64, 256, 3, 1, False, 2),     # ragged M (2700 rows), odd size, leaky
    (2, 32, 32, 512, 1, 1, True, 0),      # 1x1, two K-tiles (prologue + tail only), two column tiles
    (2, 32, 128, 256, 3, 2, False, 1),    # stride 2
    (1, 16, 1024, 256, 3, 1, False, 0),   # K = 9216
    # the 128x128 .. 64x64 tile kernel (narrow outputs / few rows)
    (8, 32, 64, 64, 3, 1, True, 1),       # trunk layer1 shape: N = 64 -> 128x64 tiles, 32-channel chunks
    (4, 32, 128, 128, 3, 1, True, 1),     # layer2 shape -> 128x128 tiles, 16-channel chunks
    (2, 16, 64, 128, 3, 2, False, 1),     # stride-2 entry conv of a stage, 64-row tiles
    (2, 16, 64, 128, 1, 2, False, 0),     # 1x1 stride-2 downsample: two K-chunks only
    (3, 7, 512, 512, 3, 1, True, 1),      # layer4 shape: odd size, ragged rows, K = 4608
    (2, 16, 48, 128, 3, 1, False, 2),     # Cin % 32 != 0: only the 128x128 tile (16-channel chunks) can take it
]


@pytest.mark.parametrize("case", BF16X3_CASES)
def test_conv_bf16x3_has_fp32_accuracy(dev, case):
    """rdpn6d_conv2d_bf16x3 (three bf16 planes per operand, six partial products on the bf16 matrix pipe) against an fp64
    convolution: its error must be no larger than the fp32-MFMA kernel's own (both are fp32-accumulated sums of products
    that are exact resp. correctly rounded), i.e. the kernel is an fp32 convolution, not a reduced-precision one.  Also: the
    three output planes re-assemble the fp32 output to 2^-24, and a second launch fed with those planes (layer chaining)
    equals a launch fed with the fp32 tensor."""
    from rdpn6d_amd import ops

    B, H, Cin, Cout, k, stride, use_res, act = case
    g = torch.Generator().manual_seed(sum(case) * 11 + 3)
    x = torch.randn(B, H, H, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g) if use_res else None
    # fp64 reference
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    y64 = y64 * sc.double() + sh.double()
    if use_res:
        y64 = y64 + res.double()
    if act == 1:
        y64 = y64.clamp(min=0)
    elif act == 2:
        y64 = torch.where(y64 > 0, y64, y64 * 0.1)
    kw = dict(stride=stride, pad=k // 2, act=act, slope=0.1)
    xd, wd, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    resd = res.to(dev) if use_res else None
    y32 = ops.conv2d_nhwc(xd, wd, scd, shd, residual=resd, **kw)
    yx3, (planes, shape) = ops.conv2d_nhwc_x3(xd, wd, scd, shd, residual=resd, want_planes=True, **kw)
    torch.cuda.synchronize()
    if use_res:  # the residual handed over as planes (what chained trunk layers do) must give the same bits
        yr = ops.conv2d_nhwc_x3(xd, wd, scd, shd, residual_planes=(ops.split_bf16x3(resd), tuple(resd.shape)), **kw)
        torch.cuda.synchronize()
        assert torch.equal(yr, yx3)
    e32 = (y32.cpu().double() - y64).abs()
    ex3 = (yx3.cpu().double() - y64).abs()
    scale = y64.abs().max().item()
    print(f"{case}: max err vs fp64  fp32-MFMA {e32.max().item():.3e}  bf16x3 {ex3.max().item():.3e}  "
          f"(rms {e32.pow(2).mean().sqrt().item():.2e} / {ex3.pow(2).mean().sqrt().item():.2e}, |y|max {scale:.2f})")
    assert ex3.max().item() <= 1.5 * e32.max().item() + 1e-7 * scale
    assert ex3.pow(2).mean().sqrt().item() <= 1.5 * e32.pow(2).mean().sqrt().item() + 1e-8 * scale
    # the planes of the output
    n = yx3.numel()
    back = planes[:, :n].float().sum(0).reshape(shape)
    assert (back - yx3).abs().max().item() <= 2.0 ** -23 * scale
    if Cout % 32 == 0 and k == 3 and stride == 1:
        # chain: y -> next conv, once from the fp32 tensor (split on the fly) and once from the planes the kernel wrote
        w2 = (torch.randn(256, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5).to(dev)
        a1 = ops.conv2d_nhwc_x3(yx3, w2, pad=1)
        a2 = ops.conv2d_nhwc_x3((planes, shape), w2, pad=1)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2)


def test_fp32_plan_fast_forms_match_the_fp32_mfma_path(golden_setup, truth, dev):
    """fp32 mode: the wide layers leave the fp32 MFMA pipe for an fp32-ACCURATE form on the 16-bit pipe - h2 (two fp16 planes,
    cfg.TEST.FP16X2, default: the whole network from ONE crop on, tile kernels at small batches, the 256x256 kernel when the batch
    fills the chip) or bf16x3 (three bf16 planes: the head from two crops on, the ResNet trunk from B = 16) - and
    cfg.TEST.BF16X3 = False keeps everything on the fp32 MFMA.  All three are fp32 evaluations of the same network: against the
    fp64 oracle each fast form must be no further away than the fp32-MFMA path (x1.25 + noise floor), and close to it."""
    models, t, gold = golden_setup
    model = models["mul"]
    assert model.plan(4, dev).fast == "h2"
    assert model.plan(4, dev).x3_launches == 57 and model.plan(4, dev).pnp_h2 and model.plan(4, dev).x3_trunk  # h2: the whole network on its tile kernels from one crop on
    assert model.plan(1, dev).x3_launches == 57
    rep = torch.arange(16, device=dev) % 4
    t16 = {k: (v[rep].contiguous() if v.shape[0] == 4 else v) for k, v in t.items()}
    tcfg = model.cfg.TEST
    outs = {}
    try:
        for mode in ("h2", "x3", "none"):
            tcfg.BF16X3, tcfg.FP16X2 = mode != "none", mode == "h2"
            model._plans.clear()
            # head: six 3x3 layers (the ConvTranspose phases have a quarter of the rows: fp32 MFMA + a split pass at this
            # batch); trunk: 32 block convolutions + 3 down-sampling ones
            plan = model.plan(16, dev)
            assert plan.fast == (None if mode == "none" else mode)
            # h2 additionally keeps the point-wise fusion branch (4 convolutions), the ConvTranspose phases (4, tile kernel) and the 1x1
            # output convolution in its format: no fp32 copies / split passes in between; + the one-pixel convolution of the folded
            # global-max half of the ConvTranspose input (cfg.TEST.FOLD_GLOBAL_MAX)
            # ... + ConvPnPNet's three convolutions and three FC layers (cfg.TEST.PNP_H2)
            # (the 1x1 output convolution rides in the last 3x3 layer's epilogue when that layer runs on the 256x256 kernel: one launch less)
            assert plan.x3_launches == {"h2": 6 + 35 + 4 + 4 + 1 + 1 + 6 - int(plan.fused_out), "x3": 6 + 35, "none": 0}[mode] and plan.x3_trunk == (mode != "none")
            assert plan.h2_pointwise == (mode == "h2")
            o = _run(model, t16)
            assert not plan.range_exceeded(wait=True)
            outs[mode] = {k: o[k].clone().cpu().double() for k in ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans")}
    finally:
        tcfg.BF16X3, tcfg.FP16X2 = True, True
        model._plans.clear()
    for mode in ("h2", "x3"):
        for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
            exact = truth["mul"][k].double()
            e3 = (outs[mode][k][:4] - exact).abs().max().item()
            e1 = (outs["none"][k][:4] - exact).abs().max().item()
            dd = (outs[mode][k] - outs["none"][k]).abs().max().item()
            print(f"{k}: vs fp64  {mode} path {e3:.3e}  fp32-MFMA path {e1:.3e}   between the two {dd:.3e}")
            assert e3 <= 1.25 * e1 + 1e-5 and dd <= 2.0 * e1 + 1e-5, (mode, k)
        for k in ("rot", "trans"):
            exact = truth["mul"][k].double()
            e3 = max(_rel(outs[mode][k][i].numpy(), exact[i].numpy()) for i in range(4))
            e1 = max(_rel(outs["none"][k][i].numpy(), exact[i].numpy()) for i in range(4))
            print(f"{k}: worst rel err vs fp64  {mode} path {e3:.3e}  fp32-MFMA path {e1:.3e}")
            assert e3 <= 2.0 * max(e1, 1e-4), (mode, k)


def test_conv_bf16x3_error_bound_under_cancellation(dev):
    """fp32-class error BOUND, not just typical error: with products that cancel almost completely (inputs and weights
    built in +/- pairs plus a small signal; magnitudes spread over 2^-10 .. 2^10) the bf16x3 result must stay within
    c * 2^-24 * sum|a||b| of the fp64 result - the a-priori bound of an fp32 dot product of length K = 2304 (c <= sqrt(K) = 48
    with random roundings) - and be no worse than the fp32-MFMA kernel's fmaf chain (measured 6.7 vs 11.9)."""
    from rdpn6d_amd import ops

    g = torch.Generator().manual_seed(99)
    B, H, C, N = 2, 32, 256, 256
    mag = torch.exp2(torch.randint(-10, 11, (B, H, H, C // 2), generator=g).float())
    half = torch.randn(B, H, H, C // 2, generator=g) * mag
    x = torch.cat([half, half], dim=-1)  # channel c and c + C/2 carry the same value ...
    wh = torch.randn(N, C // 2, 3, 3, generator=g) / 48.0
    w = torch.cat([wh, -wh], dim=1) + torch.randn(N, C, 3, 3, generator=g) * 1e-6  # ... and opposite weights (+ a tiny signal)
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    bound = torch.nn.functional.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs(), padding=1).permute(0, 2, 3, 1) * 2.0 ** -24
    y3 = ops.conv2d_nhwc_x3(x.to(dev), w.to(dev), pad=1).cpu().double()
    y1 = ops.conv2d_nhwc(x.to(dev), w.to(dev), pad=1).cpu().double()
    r3 = ((y3 - y64).abs() / bound).max().item()
    r1 = ((y1 - y64).abs() / bound).max().item()
    print(f"cancellation test: |err| / (2^-24 sum|a||b|): bf16x3 {r3:.3f}  fp32-MFMA {r1:.3f};  |y|max {y64.abs().max().item():.3e} vs sum|a||b| max {bound.max().item() * 2 ** 24:.3e}")
    assert r3 <= 1.25 * r1 and r3 <= 48.0 and r1 <= 48.0
    assert y64.abs().max().item() < 1e-2 * bound.max().item() * 2 ** 24  # the case really cancels (>= 100x)


def test_transpose_and_gradient_finite_check(dev):
    """Two small round-5 entry points on their own.  rdpn6d_transpose_rc_f32: y[b][c][r] = x[b][r][c] for sizes that are and are not
    multiples of the 32 x 32 tile (the training step uses it for the ConvPnPNet map in fc1's NCHW-flatten order, conv_pnp_net.py:151).
    rdpn6d_grad_nonfinite_f32: flag = 1 exactly when one element - at EVERY position of a 16-byte vector, and in the scalar tail - is
    +Inf, -Inf or NaN; large finite values and denormals leave it 0; a set flag is cleared by the next clean call."""
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _ptr

    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(17)
    for B, R, C in ((32, 64, 128), (3, 33, 70), (1, 1, 5), (2, 100, 1)):
        x = torch.randn(B, R, C, generator=g).to(dev)
        y = torch.full((B, C, R), 7.0, device=dev)
        _lib.check(lib.rdpn6d_transpose_rc_f32(_ptr(x), B, R, C, _ptr(y), st))
        assert torch.equal(y, x.transpose(1, 2).contiguous()), (B, R, C)
    flag = torch.full((1,), 5, dtype=torch.int32, device=dev)
    for n in (4096, 4099, 7, 1 << 20):
        base = torch.randn(n, generator=g).to(dev)
        base[0], base[n // 2] = 3.0e38, 1e-42  # finite extremes
        _lib.check(lib.rdpn6d_grad_nonfinite_f32(_ptr(base), n, _ptr(flag), st))
        assert int(flag.item()) == 0, n
        for pos in sorted({0, 1, 2, 3, n // 2 + 1, n - 1, n - 2, n - 3}):
            if not 0 <= pos < n:
                continue
            for bad in (float("inf"), float("-inf"), float("nan")):
                t = base.clone()
                t[pos] = bad
                _lib.check(lib.rdpn6d_grad_nonfinite_f32(_ptr(t), n, _ptr(flag), st))
                assert int(flag.item()) == 1, (n, pos, bad)
        _lib.check(lib.rdpn6d_grad_nonfinite_f32(_ptr(base), n, _ptr(flag), st))
        assert int(flag.item()) == 0, n


@pytest.mark.parametrize("t", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 8, 512, 4), (3, 5, 64, 2), (1, 10, 2048, 4), (2, 4, 36, 4)])
def test_upsample_bilinear_backward_vs_autograd(dev, t, case):
    """rdpn6d_upsample_bilinear_backward_*: the gradient of nn.UpsamplingBilinear2d(scale_factor) (align_corners=True,
    resnet_backbone.py:280) against autograd on the same stored gradient - the vector form (channel counts that are a multiple of 4 / 8)
    and the scalar one (36 channels in 16 bits)."""
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _ptr

    lib = _lib.load()
    B, H, C, f = case
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[t]
    g = torch.Generator().manual_seed(sum(case))
    dy = torch.randn(B, H * f, H * f, C, generator=g).to(dev).to(dt)
    dx = torch.full((B, H, H, C), 9.0, device=dev, dtype=dt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(getattr(lib, f"rdpn6d_upsample_bilinear_backward_{t}")(_ptr(dy), B, H, H, C, f, _ptr(dx), st))
    x = torch.zeros(B, C, H, H, dtype=torch.float64, requires_grad=True)
    F.interpolate(x, scale_factor=f, mode="bilinear", align_corners=True).backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    err = (dx.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= (2e-6 if t == "f32" else 2.0 ** -8 if t == "bf16" else 2.0 ** -10), err
