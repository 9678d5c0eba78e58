"""Seeded inputs of the PM_LOSS_SYM cases (shared by tools/oracle/gen_pm_sym_golden.py, which runs the REAL reference on
them, and by the parity tests).  Data only - nothing here comes from the reference's sources."""
import math

import numpy as np

B, NPTS = 16, 64


def _axis_rot(axis, deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    t = math.radians(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * (K @ K)


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def sym_sets():
    """The symmetry tables cycled over the batch: none, one rotation handed over as a bare 3x3 (pose_utils.py:441-442),
    a discrete 4-fold axis (3 rotations), a 'continuous' axis sampled every 10 degrees (35 rotations), two axes."""
    z180 = _axis_rot((0, 0, 1), 180).astype(np.float32)
    z4 = np.stack([_axis_rot((0, 0, 1), d) for d in (90, 180, 270)]).astype(np.float32)
    y36 = np.stack([_axis_rot((0, 1, 0), d) for d in range(10, 360, 10)]).astype(np.float32)
    two = np.stack([_axis_rot((1, 0, 0), 180), _axis_rot((0, 1, 0), 180), _axis_rot((0, 0, 1), 180)]).astype(np.float32)
    return [None, z180, z4, y36, two]


def make_case(seed=7):
    rng = np.random.default_rng(seed)
    sets = sym_sets()
    gt = np.stack([_rand_rot(rng) for _ in range(B)])
    sym_infos, pred = [], []
    for i in range(B):
        s = sets[i % len(sets)]
        sym_infos.append(s)
        if i == 6:  # prediction == target: the incumbent must survive (strict '<')
            pred.append(gt[i].copy())
            continue
        base = gt[i]
        if s is not None and i % 2 == 1:  # prediction near a symmetric equivalent of the target
            ss = s.reshape(-1, 3, 3)
            base = gt[i] @ ss[int(rng.integers(ss.shape[0]))].astype(np.float64)
        noise = _axis_rot(rng.normal(size=3), float(rng.uniform(2.0, 25.0)))
        pred.append(noise @ base)
    pred = np.stack(pred)
    points = rng.uniform(-0.08, 0.08, size=(B, NPTS, 3))
    extents = rng.uniform(0.05, 0.25, size=(B, 3))
    f = np.float32
    return dict(pred_rots=pred.astype(f), gt_rots=gt.astype(f), points=points.astype(f), extents=extents.astype(f),
                sym_infos=sym_infos)


def pack_sym(sym_infos):
    kmax = max([0] + [s.reshape(-1, 9).shape[0] for s in sym_infos if s is not None])
    tab = np.zeros((len(sym_infos), max(kmax, 1), 9), np.float32)
    cnt = np.zeros(len(sym_infos), np.int32)
    for i, s in enumerate(sym_infos):
        if s is not None:
            m = s.reshape(-1, 9)
            tab[i, :m.shape[0]] = m
            cnt[i] = m.shape[0]
    return tab, cnt, kmax
