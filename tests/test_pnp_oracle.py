"""The C 2D-3D RANSAC-PnP oracle (oracle/pnp_oracle.c) against analytic ground truth (known pose + pixel noise + outliers).
Parity with the reference is UNPINNED (the reference calls cv2.solvePnPRansac; cv2 is absent): the oracle is validated against
geometry here and the HIP kernel is held bit-exact to the oracle (tests/test_gpu_pnp.py)."""
import ctypes

import numpy as np
import pytest

from tests.pnp_cases import make_pnp_case
from tests.ransac_cases import pose_errors

P = ctypes.c_void_p


def run_pnp_oracle(lib, c, net_pose=None, reproj_thr=3.0, iters=100, conf=0.99, seed=7, mode=0, max_t_diff=1.0, minimal="p3p"):
    """minimal: "p3p" (sets of 4, Gauss-Newton refit) | "epnp" (sets of 5, EPnP refit: cv2.SOLVEPNP_EPNP's structure)"""
    B, HW = c["B"], c["HW"]
    pose, nin = np.zeros((B, 12), np.float32), np.zeros(B, np.int32)
    msk, best = np.zeros((B, HW), np.uint8), np.zeros(B, np.int32)
    f = lib.oracle_ransac_pnp_ex
    f.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_uint, ctypes.c_int,
                  ctypes.c_float, ctypes.c_int, P, P, P, P]
    f.restype = None
    keep = [np.ascontiguousarray(c[k]) for k in ("image_points", "model_points", "counts", "cams")]
    npz = np.ascontiguousarray(net_pose, dtype=np.float32) if net_pose is not None else None
    f(*[k.ctypes.data_as(P) for k in keep], npz.ctypes.data_as(P) if npz is not None else None, B, HW, reproj_thr, iters, conf, seed, mode,
      max_t_diff, {"p3p": 0, "epnp": 1}[minimal], pose.ctypes.data_as(P), nin.ctypes.data_as(P), msk.ctypes.data_as(P), best.ctypes.data_as(P))
    return pose, nin, msk, best


@pytest.mark.parametrize("outliers", [0.0, 0.3, 0.6])
def test_pnp_recovers_known_pose(oracle_lib, outliers):
    c = make_pnp_case(B=4, outliers=outliers, seed=int(outliers * 10))
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c)
    for b in range(c["B"]):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        n = int(c["counts"][b])
        assert best[b] >= 0 and re < 1.0 and te < 0.02, (b, re, te)  # 1 px noise on a 5-25 cm object at ~1 m: depth is the weak direction
        clean = c["clean"][b, :n]
        # the mask is the consensus set of the WINNING MINIMAL hypothesis (as in cv2.solvePnPRansac): a 4-point pose under 1 px
        # noise explains most, not all, clean correspondences within 3 px; the refit on them is what makes the pose accurate
        assert msk[b, :n][clean].mean() > 0.6
        assert not (~clean).any() or msk[b, :n][~clean].mean() < 0.02
        assert nin[b] == msk[b].sum() and msk[b, n:].sum() == 0


def test_pnp_noise_free_is_exact_and_minimal_counts(oracle_lib):
    c = make_pnp_case(B=3, n=[4, 5, 200], noise_px=0.0, outliers=0.0, seed=4)
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c)
    for b in range(3):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert re < 0.05 and te < 1e-5, (b, re, te)  # (fp32 pixel coordinates: 3e-5 px)
        assert nin[b] == c["counts"][b]
    c = make_pnp_case(B=2, n=[3, 0], seed=5)   # fewer than 4 correspondences: the -100 sentinel (gdrn_evaluator.py:391-392)
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c)
    assert (pose == -100).all() and (nin == 0).all() and (best == -1).all()


def test_pnp_network_initialised_variants(oracle_lib):
    """process_net_and_pnp (gdrn_evaluator.py:187-314): the learned pose as hypothesis 0 of a 20-iteration RANSAC (mode 1), or as the
    start of a plain iterative least-squares solve (mode 2); the network pose survives below 4 points; its translation is kept when
    the solved one moved by more than max_t_diff."""
    rng = np.random.default_rng(0)

    def net_of(case):  # a good but not exact network pose
        net = np.zeros((case["B"], 12), np.float32)
        for b in range(case["B"]):
            net[b, :9] = case["R"][b].reshape(-1)
            net[b, 9:] = case["t"][b] + rng.standard_normal(3) * 0.01
        return net

    # mode 2 is a plain least-squares solve over ALL points (no outlier rejection): it gets the outlier-free case
    for mode, iters, case in ((1, 20, make_pnp_case(B=4, outliers=0.2, seed=8)), (2, 1, make_pnp_case(B=4, outliers=0.0, seed=8))):
        pose, nin, msk, best = run_pnp_oracle(oracle_lib, case, net_pose=net_of(case), iters=iters, mode=mode)
        for b in range(4):
            re, te = pose_errors(pose[b], case["R"][b], case["t"][b])
            assert re < 2.0 and te < 0.02, (mode, b, re, te)
    c2 = make_pnp_case(B=2, n=[2, 300], outliers=0.0, seed=9)
    net2 = np.tile(np.concatenate([np.eye(3).reshape(-1), [0, 0, 1.0]]).astype(np.float32), (2, 1))
    net2[1, 9:] = c2["t"][1] + np.array([0.0, 0.0, 5.0])  # network translation 5 m off: the guard keeps it
    net2[1, :9] = c2["R"][1].reshape(-1)
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c2, net_pose=net2, iters=20, mode=1)
    assert np.array_equal(pose[0], net2[0])  # below 4 points: the network pose
    assert np.allclose(pose[1, 9:], net2[1, 9:]) and pose_errors(pose[1], c2["R"][1], c2["t"][1])[0] < 1.0


def test_pnp_deterministic_and_seed_dependent(oracle_lib):
    c = make_pnp_case(B=2, outliers=0.5, seed=11)
    a, b, d = run_pnp_oracle(oracle_lib, c, seed=1), run_pnp_oracle(oracle_lib, c, seed=1), run_pnp_oracle(oracle_lib, c, seed=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[3], d[3]) or not np.array_equal(a[2], d[2])
    assert (a[3] >= 0).all() and (a[3] < 100).all()


# ----------------------------------------------------------------------------------------------------------------- EPnP minimal solver
def test_epnp_alone_is_exact_on_exact_data(oracle_lib):
    """the EPnP restatement (control points, barycentric coordinates, null space of M^T M, betas, Horn) on noise-free correspondences:
    five points - the RANSAC's minimal set - and two hundred give the pose back to fp32 pixel precision"""
    c = make_pnp_case(B=3, n=[5, 6, 200], noise_px=0.0, outliers=0.0, seed=4)
    oracle_lib.oracle_epnp.restype = ctypes.c_int
    for b in range(3):
        R, t = np.zeros(9), np.zeros(3)
        ip, mp, cam = (np.ascontiguousarray(c[k][b]) for k in ("image_points", "model_points", "cams"))
        ok = oracle_lib.oracle_epnp(ip.ctypes.data_as(P), mp.ctypes.data_as(P), int(c["counts"][b]), cam.ctypes.data_as(P), R.ctypes.data_as(P),
                                    t.ctypes.data_as(P))
        re, te = pose_errors(np.concatenate([R, t]).astype(np.float32), c["R"][b], c["t"][b])
        assert ok == 1 and re < 0.05 and te < 1e-5 and abs(np.linalg.det(R.reshape(3, 3)) - 1.0) < 1e-12, (b, ok, re, te)
    # a planar point set has no barycentric frame: refused, not a wrong pose
    flat = np.ascontiguousarray(c["model_points"][2]).copy()
    flat[:, 2] = 0.0
    assert oracle_lib.oracle_epnp(ip.ctypes.data_as(P), flat.ctypes.data_as(P), 200, cam.ctypes.data_as(P), R.ctypes.data_as(P), t.ctypes.data_as(P)) == 0


@pytest.mark.parametrize("outliers", [0.0, 0.3, 0.5])
def test_epnp_ransac_recovers_known_pose_like_p3p(oracle_lib, outliers):
    """cfg.TEST.PNP_MINIMAL = "epnp" - five-point minimal sets, the call site's 100 iterations / 3 px / 0.99, EPnP refit
    on the inliers (lib/pysixd/misc.py:170-179) - recovers the analytic pose at 0 - 50 % outliers, with an inlier set comparable to the
    P3P + 1 solver's on the same data (both are consensus sets of ONE minimal model under 1 px noise)."""
    c = make_pnp_case(B=4, outliers=outliers, seed=int(outliers * 10))
    pe, ne, me, be = run_pnp_oracle(oracle_lib, c, minimal="epnp")
    pp, np_, mp_, bp = run_pnp_oracle(oracle_lib, c, minimal="p3p")
    for b in range(c["B"]):
        n = int(c["counts"][b])
        clean = c["clean"][b, :n]
        for nm, pose, msk, nin, best in (("epnp", pe, me, ne, be), ("p3p", pp, mp_, np_, bp)):
            re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
            assert best[b] >= 0 and re < 1.0 and te < 0.02, (nm, b, re, te)
            assert msk[b, :n][clean].mean() > 0.6 and (not (~clean).any() or msk[b, :n][~clean].mean() < 0.02), (nm, b)
            assert nin[b] == msk[b].sum() and msk[b, n:].sum() == 0
        both = (me[b, :n] & mp_[b, :n]).sum() / max(1, (me[b, :n] | mp_[b, :n]).sum())
        assert both > 0.6, (b, both)  # the two consensus sets overlap (intersection over union)


def test_epnp_minimal_counts_and_determinism(oracle_lib):
    """n = 4: EPnP's null space is four-dimensional there - the crop takes the P3P + 1 path (exact); n = 5: the one minimal set;
    below 4 the sentinel; same seed same answer, another seed another winner"""
    c = make_pnp_case(B=3, n=[4, 5, 200], noise_px=0.0, outliers=0.0, seed=4)
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c, minimal="epnp")
    for b in range(3):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert re < 0.05 and te < 1e-5 and nin[b] == c["counts"][b], (b, re, te, nin[b])
    c = make_pnp_case(B=2, n=[3, 0], seed=5)
    pose, nin, msk, best = run_pnp_oracle(oracle_lib, c, minimal="epnp")
    assert (pose == -100).all() and (nin == 0).all() and (best == -1).all()
    c = make_pnp_case(B=2, outliers=0.4, seed=11)
    a, b, d = (run_pnp_oracle(oracle_lib, c, seed=s_, minimal="epnp") for s_ in (1, 1, 2))
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[3], d[3]) or not np.array_equal(a[2], d[2])
