"""The C RANSAC/Kabsch oracle against analytic ground truth (known pose + noise + outliers + holes).
Parity of this path with the reference is UNPINNED (no reference implementation; cv2 absent): the
oracle is validated against geometry here and the HIP kernel is held bit-exact to the oracle."""
import ctypes

import numpy as np
import pytest

from tests.ransac_cases import make_case, pose_errors

P = ctypes.c_void_p


def run_oracle(lib, c, mask_thr=0.5, inlier_thr=0.01, iters=100, conf=0.99, seed=7):
    B, HW, K = c["B"], c["HW"], c["K"]
    pose = np.zeros((B, 12), np.float32)
    nin = np.zeros(B, np.int32)
    msk = np.zeros((B, HW), np.uint8)
    best = np.zeros(B, np.int32)
    f = lib.oracle_ransac_kabsch
    f.argtypes = [P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                  ctypes.c_int, ctypes.c_float, ctypes.c_uint, P, P, P, P]
    f.restype = None
    a = lambda x: np.ascontiguousarray(x).ctypes.data_as(P)
    keep = [np.ascontiguousarray(c[k]) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")]
    f(*[k.ctypes.data_as(P) for k in keep], B, HW, K, mask_thr, inlier_thr, iters, conf, seed, a(pose), a(nin), a(msk), a(best))
    return pose, nin, msk, best


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
