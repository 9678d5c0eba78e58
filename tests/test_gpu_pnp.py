"""2D-3D RANSAC-PnP on the GPU (rdpn6d_ransac_pnp_f32, rows A9 / A10): inlier masks, counts and the winning hypothesis BIT-EXACT
against the C oracle (oracle/pnp_oracle.c) under a fixed seed, the refit pose to 1e-5, analytic ground truth recovered; the
network-initialised variants of process_net_and_pnp; and cfg.TEST.PNP_TYPE = "ransac_pnp" inside GDRN.forward = selection
(row A8) + this solve on the model's own maps.  Parity with cv2.solvePnPRansac itself is UNPINNED (cv2 is absent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(c, net=None):
    dev = torch.device("cuda:0")
    g = {k: torch.from_numpy(np.ascontiguousarray(c[k])).to(dev) for k in ("image_points", "model_points", "counts", "cams")}
    return g, (torch.from_numpy(net).to(dev) if net is not None else None)


@pytest.mark.parametrize("outliers,n", [(0.0, 1500), (0.3, 1500), (0.6, 1500), (0.4, 4096), (0.3, 40)])
def test_pnp_bit_exact_vs_oracle_and_ground_truth(oracle_lib, outliers, n):
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.ransac_cases import pose_errors
    from tests.test_pnp_oracle import run_pnp_oracle

    c = make_pnp_case(B=5, n=n, outliers=outliers, seed=int(outliers * 10) + n)
    g, _ = _dev(c)
    for seed in (1, 99):
        po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=seed)
        pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(5, 3, 3), seed=seed)
        torch.cuda.synchronize()
        assert np.array_equal(best.cpu().numpy(), bo), (best.cpu().numpy(), bo)
        assert np.array_equal(nin.cpu().numpy(), ni)
        assert np.array_equal(msk.cpu().numpy(), mo)
        assert np.abs(pose.cpu().numpy() - po).max() < 1e-5
        for b in range(c["B"]):
            re, te = pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])
            # (a small object under 1 px noise: ~0.5 deg / ~5 mm is the solve's own accuracy at these sizes; heavier outlier ratios leave fewer inliers)
            assert re < (2.0 if n > 100 else 5.0) and te < (0.03 if n > 100 else 0.1), (b, re, te)


def test_pnp_full_batch_outlier_sweep_b64(oracle_lib):
    """C4's stress shape for the reference's solver: 64 crops in one launch, up to 1 600 correspondences each, outliers 0 .. 70 %"""
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.ransac_cases import pose_errors
    from tests.test_pnp_oracle import run_pnp_oracle

    B = 64
    ratios = np.linspace(0.0, 0.7, B)
    c = make_pnp_case(B=B, n=1600, outliers=ratios, seed=5)
    g, _ = _dev(c)
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=3)
    pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(B, 3, 3), seed=3)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(nin.cpu().numpy(), ni) and np.array_equal(msk.cpu().numpy(), mo)
    real = ni >= 50
    assert np.abs(pose.cpu().numpy() - po)[real].max() < 1e-5
    ok = sum(pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])[0] < 1.0 for b in range(B))
    print(f"2D-3D RANSAC-PnP, 64 crops, outliers 0..70 %: pose recovered (< 1 deg) on {ok} / {B}")
    assert ok >= 56
    for b in range(B):
        if ratios[b] <= 0.5:
            re, te = pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])
            assert re < 2.0 and te < 0.03, (b, ratios[b], re, te)


def test_pnp_sentinel_and_network_initialised_variants(oracle_lib):
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.test_pnp_oracle import run_pnp_oracle

    c = make_pnp_case(B=4, n=[3, 0, 4, 300], noise_px=0.0, outliers=0.0, seed=5)
    g, _ = _dev(c)
    pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(4, 3, 3), seed=2)
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=2)
    torch.cuda.synchronize()
    assert (pose[:2].cpu().numpy() == -100).all() and (nin[:2].cpu().numpy() == 0).all() and (best[:2].cpu().numpy() == -1).all()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(msk.cpu().numpy(), mo) and np.abs(pose.cpu().numpy() - po).max() < 1e-5
    rng = np.random.default_rng(0)
    for mode, iters, outl in (("ransac", 20, 0.2), ("iter", 1, 0.0)):
        c = make_pnp_case(B=6, n=[2, 900, 900, 900, 900, 900], outliers=outl, seed=8)
        net = np.zeros((6, 12), np.float32)
        for b in range(6):
            net[b, :9], net[b, 9:] = c["R"][b].reshape(-1), c["t"][b] + rng.standard_normal(3) * 0.01
        net[5, 9:] += np.array([0, 0, 5.0], np.float32)  # a network translation 5 m off: the 1 m guard keeps it
        g, nd = _dev(c, net)
        po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, net_pose=net, iters=iters, mode=1 if mode == "ransac" else 2, seed=4)
        pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(6, 3, 3), iters=iters, seed=4,
                                              net_pose=nd, net_mode=mode)
        torch.cuda.synchronize()
        assert np.array_equal(pose[0].cpu().numpy(), net[0])  # fewer than 4 correspondences: the network pose
        assert np.allclose(pose[5, 9:].cpu().numpy(), net[5, 9:])
        assert np.array_equal(nin.cpu().numpy(), ni) and np.array_equal(msk.cpu().numpy(), mo) and np.array_equal(best.cpu().numpy(), bo)
        assert np.abs(pose.cpu().numpy() - po).max() < 1e-5


@pytest.mark.parametrize("pnp_type", ["ransac_pnp", "net_ransac_pnp", "net_iter_pnp"])
def test_model_pnp_type_2d3d_inside_forward(oracle_lib, golden_dir, pnp_type):
    """cfg.TEST.USE_PNP with the reference's own PNP_TYPE values (gdrn_evaluator.py:136-145): inside GDRN.forward the maps go through
    the A8 selection and the 2D-3D solve; the result equals the numpy selection oracle + the C PnP oracle run on those same maps."""
    import os

    from oracle import select_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from tests.c1w_cases import c1w_state_dict
    from tests.test_pnp_oracle import run_pnp_oracle

    dev = torch.device("cuda:0")
    cfg = gdrn_base_cfg(mask_attention="none", device="cuda")
    cfg.TEST.USE_PNP, cfg.TEST.PNP_TYPE, cfg.TEST.PNP_SEED = True, pnp_type, 3
    model, _ = build_model_optimizer(cfg)
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    sd = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    inp = synth.make_inputs(4, seed=36)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    # per-crop image sizes as the reference's batch carries them (engine_utils.py:71 "im_H", "im_W"; gdrn_evaluator.py:346-347): crop 1
    # comes from a 540 x 720 image (T-LESS), the others from 480 x 640
    im_H, im_W = torch.tensor([480.0, 540.0, 480.0, 480.0]), torch.tensor([640.0, 720.0, 640.0, 640.0])
    kw = dict(roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
              roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
    with torch.no_grad():
        with pytest.raises(ValueError, match="im_H"):  # no silent 480 x 640
            model(t["roi_img"], **kw)
        o = model(t["roi_img"], im_H=im_H.to(dev), im_W=im_W, **kw)
    torch.cuda.synchronize()
    maps = torch.cat([o["mask"], o["coor_x"], o["coor_y"], o["coor_z"]], 1).cpu().numpy()
    nm = select_oracle.out_mask_l1(maps[:, :1])
    ip, mp, cnt = np.zeros((4, 4096, 2), np.float32), np.zeros((4, 4096, 3), np.float32), np.zeros(4, np.int32)
    for b in range(4):
        a, m, _ = select_oracle.select_correspondences(nm[b, 0], maps[b, 1:4].transpose(1, 2, 0), inp["roi_coord_2d"][b, 3:5].transpose(1, 2, 0),
                                                       int(im_H[b]), int(im_W[b]), inp["roi_extent"][b], 0.5)
        cnt[b] = len(a)
        ip[b, :len(a)], mp[b, :len(a)] = a, m
    assert np.array_equal(o["pnp_num_points"].cpu().numpy(), cnt) and cnt.min() >= 4
    c = dict(image_points=ip, model_points=mp, counts=cnt, cams=inp["roi_cam"].reshape(4, 9), B=4, HW=4096)
    net = np.concatenate([o["rot"].cpu().numpy().reshape(4, 9), o["trans"].cpu().numpy()], 1).astype(np.float32)
    mode = {"ransac_pnp": 0, "net_ransac_pnp": 1, "net_iter_pnp": 2}[pnp_type]
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, net_pose=net if mode else None, iters=20 if mode == 1 else 100, mode=mode, seed=3)
    assert np.array_equal(o["pnp_num_inliers"].cpu().numpy(), ni) and np.array_equal(o["pnp_inlier_mask"].cpu().numpy(), mo)
    assert np.abs(o["pnp_pose"].cpu().numpy() - po).max() < 1e-4
    assert o["pnp_pose"].shape == (4, 12)
    if pnp_type == "ransac_pnp":
        # cfg.TEST.IM_H / IM_W (both set) serve a loop whose images all have one size; a mask type whose normalisation the
        # selection kernel does not implement (get_out_mask's sigmoid branch, engine_utils.py:130-132) raises instead of using min-max
        cfg.TEST.IM_H, cfg.TEST.IM_W = 480, 640
        with torch.no_grad():
            o2 = model(t["roi_img"], **kw)
            o3 = model(t["roi_img"], im_H=480, im_W=[640] * 4, **kw)
        for k in ("pnp_pose", "pnp_num_points", "pnp_inlier_mask"):
            assert torch.equal(o2[k], o3[k])
        assert not torch.equal(o2["pnp_pose"][1], o["pnp_pose"][1]) and torch.equal(o2["pnp_num_points"], o["pnp_num_points"])
        # round 5: get_out_mask's sigmoid branch (engine_utils.py:130-132) is built.  Switching the mask type on a live model re-builds the
        # plan; the selection then keeps sigmoid(mask) > 0.5 (& the |xyz| filter) instead of the min-max normalised mask > 0.5
        from oracle import select_oracle

        cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = "BCE"
        with torch.no_grad():
            o4 = model(t["roi_img"], **kw)
        torch.cuda.synchronize()
        assert model.plan(4, t["roi_img"].device).mask_type == 1
        nm = select_oracle.out_mask(o4["mask"].cpu().numpy(), "BCE")
        for b in range(4):
            xyz = np.stack([o4[k][b, 0].cpu().numpy() for k in ("coor_x", "coor_y", "coor_z")], -1)
            _, _, sel = select_oracle.select_correspondences(nm[b, 0], xyz, t["roi_coord_2d"][b, 3:5].permute(1, 2, 0).cpu().numpy(), 480, 640,
                                                             t["roi_extent"][b].cpu().numpy(), 0.5)
            assert int(o4["pnp_num_points"][b]) == int(sel.sum()), (b, int(o4["pnp_num_points"][b]), int(sel.sum()))
        assert not torch.equal(o4["pnp_num_points"], o2["pnp_num_points"])
        cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = "L1"


# ------------------------------------------------------------------------------------------------------- EPnP minimal solver (round 6)
@pytest.mark.parametrize("outliers,n", [(0.0, 1500), (0.3, 1500), (0.5, 1500), (0.4, 4096), (0.3, 40)])
def test_epnp_bit_exact_vs_oracle_and_ground_truth(oracle_lib, outliers, n):
    """cfg.TEST.PNP_MINIMAL = "epnp" (VERDICT r5 item 7): five-point minimal sets solved by EPnP on one wavefront each - the 12 x 12
    Jacobi on lanes 0..11 of an LDS scratch - must give the C oracle's inlier masks, counts and winning hypothesis BIT FOR BIT; the
    EPnP refit over the inliers (block-tree sums instead of the oracle's serial ones) to 1e-5; analytic ground truth recovered."""
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.ransac_cases import pose_errors
    from tests.test_pnp_oracle import run_pnp_oracle

    c = make_pnp_case(B=5, n=n, outliers=outliers, seed=int(outliers * 10) + n)
    g, _ = _dev(c)
    for seed in (1, 99):
        po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=seed, minimal="epnp")
        pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(5, 3, 3), seed=seed, minimal="epnp")
        torch.cuda.synchronize()
        assert np.array_equal(best.cpu().numpy(), bo), (best.cpu().numpy(), bo)
        assert np.array_equal(nin.cpu().numpy(), ni)
        assert np.array_equal(msk.cpu().numpy(), mo)
        assert np.abs(pose.cpu().numpy() - po).max() < 1e-5
        errs = [pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b]) for b in range(c["B"])]
        if outliers <= 0.4:
            for b, (re, te) in enumerate(errs):
                assert re < (2.0 if n > 100 else 5.0) and te < (0.03 if n > 100 else 0.1), (b, re, te)
        else:
            # five-point sets at 50 % outliers: 3 % of the draws are clean and a clean minimal set under 1 px noise is a mediocre model -
            # the consensus winner (bit-identical to the oracle's, above) is now and then a weak one and the EPnP refit inherits it
            # (measured: 3.8 deg on one crop of five); the reference's own call has the same structure
            assert sum(re < 2.0 and te < 0.03 for re, te in errs) >= c["B"] - 1 and all(re < 8.0 and te < 0.1 for re, te in errs), errs


def test_epnp_full_batch_outlier_sweep_b64_next_to_p3p(oracle_lib):
    """64 crops in one launch, outliers 0 .. 60 %: bit-exact vs the oracle; up to 40 % outliers both minimal solvers recover the pose
    on every crop (beyond that the five-point sets get rare inside the call's 100 iterations), and their consensus sets overlap"""
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.ransac_cases import pose_errors
    from tests.test_pnp_oracle import run_pnp_oracle

    B = 64
    ratios = np.linspace(0.0, 0.6, B)
    c = make_pnp_case(B=B, n=1600, outliers=ratios, seed=5)
    g, _ = _dev(c)
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=3, minimal="epnp")
    pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(B, 3, 3), seed=3, minimal="epnp")
    p3, n3, m3, b3 = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(B, 3, 3), seed=3, minimal="p3p")
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(nin.cpu().numpy(), ni) and np.array_equal(msk.cpu().numpy(), mo)
    real = ni >= 50
    assert np.abs(pose.cpu().numpy() - po)[real].max() < 1e-5
    me, mp_ = msk.cpu().numpy().astype(bool), m3.cpu().numpy().astype(bool)
    for b in range(B):
        if ratios[b] <= 0.4:
            for nm, ps in (("epnp", pose), ("p3p", p3)):
                re, te = pose_errors(ps[b].cpu().numpy(), c["R"][b], c["t"][b])
                assert re < 2.0 and te < 0.03, (nm, b, ratios[b], re, te)
            iou = (me[b] & mp_[b]).sum() / max(1, (me[b] | mp_[b]).sum())
            assert iou > 0.35, (b, iou)  # (two consensus sets of ONE minimal model each under 1 px noise: measured 0.47 .. 0.95)


def test_epnp_small_counts_and_network_initialised_variants(oracle_lib):
    """below 4 correspondences the sentinel, exactly 4 the P3P + 1 path, 5 the one minimal set; the network pose as hypothesis 0 of a
    20-iteration run (mode 1) with the EPnP models and the EPnP refit - all bit-exact vs the oracle"""
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case
    from tests.ransac_cases import pose_errors
    from tests.test_pnp_oracle import run_pnp_oracle

    c = make_pnp_case(B=5, n=[3, 0, 4, 5, 300], noise_px=0.0, outliers=0.0, seed=5)
    g, _ = _dev(c)
    pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(5, 3, 3), seed=2, minimal="epnp")
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=2, minimal="epnp")
    torch.cuda.synchronize()
    assert (pose[:2].cpu().numpy() == -100).all() and (nin[:2].cpu().numpy() == 0).all() and (best[:2].cpu().numpy() == -1).all()
    assert np.array_equal(best.cpu().numpy(), bo) and np.array_equal(msk.cpu().numpy(), mo) and np.abs(pose.cpu().numpy() - po).max() < 1e-5
    for b in (2, 3, 4):
        re, te = pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])
        assert re < 0.05 and te < 1e-4 and int(nin[b]) == int(c["counts"][b]), (b, re, te)
    rng = np.random.default_rng(0)
    c = make_pnp_case(B=6, n=[2, 900, 900, 900, 900, 900], outliers=0.2, seed=8)
    net = np.zeros((6, 12), np.float32)
    for b in range(6):
        net[b, :9], net[b, 9:] = c["R"][b].reshape(-1), c["t"][b] + rng.standard_normal(3) * 0.01
    g, nd = _dev(c, net)
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, net_pose=net, iters=20, mode=1, seed=4, minimal="epnp")
    pose, nin, msk, best = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(6, 3, 3), iters=20, seed=4,
                                          net_pose=nd, net_mode="ransac", minimal="epnp")
    torch.cuda.synchronize()
    assert np.array_equal(pose[0].cpu().numpy(), net[0])
    assert np.array_equal(nin.cpu().numpy(), ni) and np.array_equal(msk.cpu().numpy(), mo) and np.array_equal(best.cpu().numpy(), bo)
    assert np.abs(pose.cpu().numpy() - po).max() < 1e-5


def test_model_pnp_minimal_epnp_inside_forward(oracle_lib, golden_dir):
    """cfg.TEST.PNP_MINIMAL = "epnp" with PNP_TYPE = "ransac_pnp" inside GDRN.forward: the A8 selection + the EPnP RANSAC on the model's own
    maps equal the numpy selection oracle + the C oracle with minimal = epnp; an unknown value raises"""
    import os

    from oracle import select_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from tests.c1w_cases import c1w_state_dict
    from tests.test_pnp_oracle import run_pnp_oracle

    dev = torch.device("cuda:0")
    cfg = gdrn_base_cfg(mask_attention="none", device="cuda")
    cfg.TEST.USE_PNP, cfg.TEST.PNP_TYPE, cfg.TEST.PNP_SEED, cfg.TEST.PNP_MINIMAL = True, "ransac_pnp", 3, "epnp"
    cfg.TEST.IM_H, cfg.TEST.IM_W = 480, 640
    model, _ = build_model_optimizer(cfg)
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    sd = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    inp = synth.make_inputs(4, seed=36)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    kw = dict(roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
              roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
    with torch.no_grad():
        o = model(t["roi_img"], **kw)
    torch.cuda.synchronize()
    maps = torch.cat([o["mask"], o["coor_x"], o["coor_y"], o["coor_z"]], 1).cpu().numpy()
    nm = select_oracle.out_mask_l1(maps[:, :1])
    ip, mp, cnt = np.zeros((4, 4096, 2), np.float32), np.zeros((4, 4096, 3), np.float32), np.zeros(4, np.int32)
    for b in range(4):
        a, m, _ = select_oracle.select_correspondences(nm[b, 0], maps[b, 1:4].transpose(1, 2, 0), inp["roi_coord_2d"][b, 3:5].transpose(1, 2, 0),
                                                       480, 640, inp["roi_extent"][b], 0.5)
        cnt[b] = len(a)
        ip[b, :len(a)], mp[b, :len(a)] = a, m
    c = dict(image_points=ip, model_points=mp, counts=cnt, cams=inp["roi_cam"].reshape(4, 9), B=4, HW=4096)
    po, ni, mo, bo = run_pnp_oracle(oracle_lib, c, seed=3, minimal="epnp")
    assert np.array_equal(o["pnp_num_inliers"].cpu().numpy(), ni) and np.array_equal(o["pnp_inlier_mask"].cpu().numpy(), mo)
    assert np.abs(o["pnp_pose"].cpu().numpy() - po).max() < 1e-4
    model.cfg.TEST.PNP_MINIMAL = "dlt"
    with torch.no_grad(), pytest.raises(ValueError, match="PNP_MINIMAL"):
        model(t["roi_img"], **kw)


@pytest.mark.parametrize("minimal", ["p3p", "epnp"])
@pytest.mark.parametrize("B", [3, 64])
def test_pnp_split_over_workgroups_is_bit_identical(minimal, B):
    """with fewer crops than compute units a crop's hypotheses are spread over several workgroups (global scoreboard, second launch for
    scan + refit): poses, masks, counts and winners equal the one-launch form's bit for bit, both minimal solvers, per-image and full
    batches, with and without the network pose as hypothesis 0"""
    from rdpn6d_amd import ops
    from tests.pnp_cases import make_pnp_case

    c = make_pnp_case(B=B, n=[3, 4, 900] if B == 3 else 1200, outliers=0.3, seed=B)
    net = np.zeros((B, 12), np.float32)
    for b in range(B):
        net[b, :9], net[b, 9:] = c["R"][b].reshape(-1), c["t"][b] + 0.01
    g, nd = _dev(c, net)
    for kw in (dict(), dict(net_pose=nd, net_mode="ransac", iters=20)):
        a = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(B, 3, 3), seed=5, minimal=minimal, split=True, **kw)
        u = ops.ransac_pnp(g["image_points"], g["model_points"], g["counts"], g["cams"].reshape(B, 3, 3), seed=5, minimal=minimal, split=False, **kw)
        torch.cuda.synchronize()
        for x, y in zip(a, u):
            assert torch.equal(x, y)
