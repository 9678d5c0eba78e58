"""seeded inputs shared by the targets / pose-error golden generator and tests"""
import numpy as np

from tests.ransac_cases import rand_rot


def target_case(seed, K=32, side=64):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 808])))
    ext = (rng.random(3) * 0.2 + 0.05).astype(np.float32)
    v = rng.standard_normal((K + 1, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    fps = (v * 0.5 * ext)[:K].astype(np.float64)  # the loader keeps float64 anchors
    xyz = ((rng.random((side, side, 3)) - 0.5) * ext).astype(np.float32)
    yy, xx = np.mgrid[0:side, 0:side]
    bg = (xx - side / 2) ** 2 + (yy - side / 2) ** 2 > (side * 0.35) ** 2
    xyz[bg] = 0
    xyz[5, 5] = fps[3].astype(np.float32)  # a pixel (almost) on an anchor
    R = rand_rot(rng).astype(np.float32)
    return xyz, fps, R, ext


def pose_case(seed, n=600):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 909])))
    pts = (rng.random((n, 3)) - 0.5) * np.array([0.1, 0.2, 0.15])
    Rg = rand_rot(rng)
    tg = np.array([0.05, -0.03, 0.8])
    dq = rand_rot(rng)
    a = rng.random() * 0.2
    Re = Rg @ (np.eye(3) * (1 - a) + dq * a)
    u, _, vt = np.linalg.svd(Re)
    Re = u @ vt
    te = tg + rng.standard_normal(3) * 0.01
    return Re, te, Rg, tg, pts
