"""The C RANSAC/Kabsch oracle against analytic ground truth (known pose + noise + outliers + holes).
Parity of this path with the reference is UNPINNED (no reference implementation; cv2 absent): the
oracle is validated against geometry here and the HIP kernel is held bit-exact to the oracle."""
import ctypes

import numpy as np
import pytest

from tests.ransac_cases import make_case, pose_errors

P = ctypes.c_void_p


def run_oracle(lib, c, mask_thr=0.5, inlier_thr=0.01, iters=100, conf=0.99, seed=7, mask_type=0):
    """mask_type 0 L1 (min-max) | 1 BCE (sigmoid) | 2 CE (arg-max over two mask channels): ROT_HEAD.MASK_LOSS_TYPE as
    engine_utils.get_out_mask reads it"""
    B, HW, K = c["B"], c["HW"], c["K"]
    pose = np.zeros((B, 12), np.float32)
    nin = np.zeros(B, np.int32)
    msk = np.zeros((B, HW), np.uint8)
    best = np.zeros(B, np.int32)
    a = lambda x: np.ascontiguousarray(x).ctypes.data_as(P)
    keep = [np.ascontiguousarray(c[k]) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")]
    if mask_type:
        f = lib.oracle_ransac_kabsch_mt
        f.argtypes = [P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                      ctypes.c_int, ctypes.c_float, ctypes.c_uint, P, P, P, P]
        f.restype = None
        f(*[k.ctypes.data_as(P) for k in keep], B, HW, K, mask_thr, mask_type, inlier_thr, iters, conf, seed, a(pose), a(nin), a(msk), a(best))
        return pose, nin, msk, best
    f = lib.oracle_ransac_kabsch
    f.argtypes = [P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                  ctypes.c_int, ctypes.c_float, ctypes.c_uint, P, P, P, P]
    f.restype = None
    f(*[k.ctypes.data_as(P) for k in keep], B, HW, K, mask_thr, inlier_thr, iters, conf, seed, a(pose), a(nin), a(msk), a(best))
    return pose, nin, msk, best


def with_mask_type(c, mask_type):
    """the case `c` (min-max mask in channel 0) re-expressed for MASK_LOSS_TYPE BCE (1: a logit per pixel) or CE (2: two mask
    channels) such that the SAME pixels pass mask > 0.5: the solve must then be identical bit for bit"""
    o = np.asarray(c["out_nchw"], np.float32)
    B, C, HW = o.shape
    m = o[:, 0]
    nm = (m - m.min(1, keepdims=True)) / (m.max(1, keepdims=True) - m.min(1, keepdims=True))
    z = np.where(nm > np.float32(0.5), np.float32(2.0) + nm, np.float32(-2.0) - nm).astype(np.float32)  # sigmoid(z) > 0.5  <=>  nm > 0.5
    d = dict(c)
    if mask_type == 1:
        d["out_nchw"] = np.concatenate([z[:, None], o[:, 1:]], 1)
    else:
        d["out_nchw"] = np.concatenate([np.zeros_like(z)[:, None], z[:, None], o[:, 1:]], 1)
    return d


@pytest.mark.parametrize("mask_type", [1, 2])
def test_mask_loss_types_select_like_get_out_mask(oracle_lib, mask_type):
    """BCE / CE reading of the mask (engine_utils.py:130-134): with logits built so that the same pixels pass the threshold, the
    solve equals the L1 run bit for bit; flipping the sign of every logit selects the complement"""
    c = make_case(B=3, outliers=0.3, seed=5)
    want = run_oracle(oracle_lib, c)
    got = run_oracle(oracle_lib, with_mask_type(c, mask_type), mask_type=mask_type)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    neg = with_mask_type(c, mask_type)
    neg["out_nchw"] = neg["out_nchw"].copy()
    neg["out_nchw"][:, mask_type - 1] *= -1.0
    _, nin, msk, _ = run_oracle(oracle_lib, neg, mask_type=mask_type)
    assert not np.array_equal(msk, want[2])


@pytest.mark.parametrize("outliers", [0.0, 0.3, 0.6])
def test_recovers_known_pose(oracle_lib, outliers):
    c = make_case(B=4, outliers=outliers, seed=int(outliers * 10))
    pose, nin, msk, best = run_oracle(oracle_lib, c)
    for b in range(c["B"]):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert best[b] >= 0 and re < 0.5 and te < 0.002, (b, re, te)
        clean = c["clean"][b]
        # almost every clean foreground pixel is an inlier, almost no corrupted one is
        assert msk[b][clean].mean() > 0.97
        assert msk[b][~clean].mean() < 0.02
        assert nin[b] == msk[b].sum()


def test_too_few_points_gives_sentinel(oracle_lib):
    c = make_case(B=2, seed=3)
    c["out_nchw"][:, 0] = 0.0          # nothing passes the mask threshold ...
    c["out_nchw"][:, 0, 0], c["out_nchw"][:, 0, 1] = -1.0, 1.0
    c["out_nchw"][1, 0, 5] = 0.9       # ... one pixel does in crop 1
    pose, nin, msk, best = run_oracle(oracle_lib, c)
    assert (pose == -100).all() and (nin == 0).all() and (best == -1).all() and msk.sum() == 0


def test_deterministic_and_seed_dependent(oracle_lib):
    c = make_case(B=2, outliers=0.5, seed=9)
    a = run_oracle(oracle_lib, c, seed=1)
    b = run_oracle(oracle_lib, c, seed=1)
    d = run_oracle(oracle_lib, c, seed=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[3], d[3]) or not np.array_equal(a[2], d[2])
    # adaptive stop: with few outliers the winning hypothesis comes early
    e = run_oracle(oracle_lib, make_case(B=4, outliers=0.05, seed=11))
    assert (e[3] < 30).all()


# ----------------------------------------------------------------------------- network-initialised variants (A10)
def run_oracle_net(lib, c, net_pose, mode, mask_thr=0.5, inlier_thr=0.01, iters=20, conf=0.99, seed=7, max_t_diff=1.0):
    B, HW, K = c["B"], c["HW"], c["K"]
    pose = np.zeros((B, 12), np.float32)
    nin = np.zeros(B, np.int32)
    msk = np.zeros((B, HW), np.uint8)
    best = np.zeros(B, np.int32)
    f = lib.oracle_ransac_kabsch_net
    f.argtypes = [P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                  ctypes.c_float, ctypes.c_uint, ctypes.c_int, ctypes.c_float, P, P, P, P]
    f.restype = None
    a = lambda x: np.ascontiguousarray(x).ctypes.data_as(P)  # noqa: E731
    keep = [np.ascontiguousarray(c[k]) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")]
    npz = np.ascontiguousarray(net_pose, np.float32)
    f(*[k.ctypes.data_as(P) for k in keep], a(npz), B, HW, K, mask_thr, inlier_thr, iters, conf, seed, mode, max_t_diff,
      a(pose), a(nin), a(msk), a(best))
    return pose, nin, msk, best


def _gt_pose12(c):
    return np.concatenate([c["R"].reshape(c["B"], 9), c["t"]], 1).astype(np.float32)


def test_net_ransac_good_and_bad_initial_pose(oracle_lib):
    """role of process_net_and_pnp(pnp_type="ransac") (gdrn_evaluator.py:263-277): the learned pose is one more hypothesis"""
    c = make_case(B=4, outliers=0.4, seed=21)
    good = _gt_pose12(c)
    pose, nin, msk, best = run_oracle_net(oracle_lib, c, good, mode=1)
    assert (best == 0).all()                    # the exact pose wins as hypothesis 0 and stops the scan early
    for b in range(4):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert re < 0.5 and te < 0.002
    bad = good.copy()
    bad[:, :9] = np.eye(3).reshape(9)           # wrong rotation, right translation: sampled hypotheses must win
    pose, nin, msk, best = run_oracle_net(oracle_lib, c, bad, mode=1, iters=100)
    assert (best > 0).all()
    for b in range(4):
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert re < 0.5 and te < 0.002
    # plain variant, same seed: hypotheses 1.. are the same draws, so a bad initial pose changes nothing but slot 0
    p0, n0, m0, b0 = run_oracle(oracle_lib, c, inlier_thr=0.01, iters=100, seed=7)
    same = b0 > 0
    assert np.array_equal(best[same], b0[same]) and np.array_equal(msk[same], m0[same])


def test_net_variants_fallbacks(oracle_lib):
    """fewer than 3 correspondences -> the network pose (gdrn_evaluator.py:297-300); solved t further than max_t_diff from
    the network's t -> the network's t (:293-296)"""
    c = make_case(B=2, outliers=0.0, seed=3)
    net = _gt_pose12(c) + 0.25
    c2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    c2["out_nchw"][:, 0] = 0.0
    c2["out_nchw"][:, 0, 0], c2["out_nchw"][:, 0, 1] = -1.0, 1.0
    for mode in (1, 2):
        pose, nin, msk, best = run_oracle_net(oracle_lib, c2, net, mode=mode)
        assert np.array_equal(pose, net) and (nin == 0).all() and (best == -1).all()
    far = _gt_pose12(c)
    far[:, 9:] += np.array([0.0, 0.0, 1.5], np.float32)   # network t 1.5 m away from what the points say
    for mode in (1, 2):
        pose, nin, msk, best = run_oracle_net(oracle_lib, c, far, mode=mode, iters=100)
        assert np.array_equal(pose[:, 9:], far[:, 9:])     # guard fired: network translation kept
        for b in range(2):
            assert pose_errors(pose[b], c["R"][b], far[b, 9:])[0] < 1.0  # rotation still the solved one
        pose2, *_ = run_oracle_net(oracle_lib, c, far, mode=mode, iters=100, max_t_diff=2.0)
        for b in range(2):
            assert pose_errors(pose2[b], c["R"][b], c["t"][b])[1] < 0.01  # guard not triggered: solved t


def test_net_iter_is_least_squares_over_all_points(oracle_lib):
    """role of solvePnP(ITERATIVE, useExtrinsicGuess) (gdrn_evaluator.py:278-291): a least-squares fit over ALL selected
    correspondences - exact on clean data, pulled by outliers (that is what the reference's variant does too)"""
    c = make_case(B=3, outliers=0.0, noise=0.0005, seed=5)
    net = _gt_pose12(c)
    pose, nin, msk, best = run_oracle_net(oracle_lib, c, net, mode=2)
    assert (best == 0).all()
    for b in range(3):
        assert nin[b] == msk[b].sum() and nin[b] > 100      # every selected correspondence is used
        re, te = pose_errors(pose[b], c["R"][b], c["t"][b])
        assert re < 0.3 and te < 0.001
    co = make_case(B=3, outliers=0.5, seed=6)
    pose_o, *_ = run_oracle_net(oracle_lib, co, _gt_pose12(co), mode=2)
    pose_r, *_ = run_oracle_net(oracle_lib, co, _gt_pose12(co), mode=1)
    err_o = max(pose_errors(pose_o[b], co["R"][b], co["t"][b])[0] for b in range(3))
    err_r = max(pose_errors(pose_r[b], co["R"][b], co["t"][b])[0] for b in range(3))
    assert err_r < 0.5 < err_o  # RANSAC variant shrugs the outliers off, the all-points fit does not
