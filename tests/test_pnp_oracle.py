"""The C 2D-3D RANSAC-PnP oracle (oracle/pnp_oracle.c) against analytic ground truth (known pose + pixel noise + outliers).
Parity with the reference is UNPINNED (the reference calls cv2.solvePnPRansac; cv2 is absent): the oracle is validated against
geometry here and the HIP kernel is held bit-exact to the oracle (tests/test_gpu_pnp.py)."""
import ctypes

import numpy as np
import pytest

from tests.pnp_cases import make_pnp_case
from tests.ransac_cases import pose_errors

P = ctypes.c_void_p


def run_pnp_oracle(lib, c, net_pose=None, reproj_thr=3.0, iters=100, conf=0.99, seed=7, mode=0, max_t_diff=1.0):
    B, HW = c["B"], c["HW"]
    pose, nin = np.zeros((B, 12), np.float32), np.zeros(B, np.int32)
    msk, best = np.zeros((B, HW), np.uint8), np.zeros(B, np.int32)
    f = lib.oracle_ransac_pnp
    f.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_uint, ctypes.c_int,
                  ctypes.c_float, P, P, P, P]
    f.restype = None
    keep = [np.ascontiguousarray(c[k]) for k in ("image_points", "model_points", "counts", "cams")]
    npz = np.ascontiguousarray(net_pose, dtype=np.float32) if net_pose is not None else None
    f(*[k.ctypes.data_as(P) for k in keep], npz.ctypes.data_as(P) if npz is not None else None, B, HW, reproj_thr, iters, conf, seed, mode,
      max_t_diff, pose.ctypes.data_as(P), nin.ctypes.data_as(P), msk.ctypes.data_as(P), best.ctypes.data_as(P))
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
