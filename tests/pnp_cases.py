"""Synthetic 2D-3D correspondences with a KNOWN pose for the RANSAC-PnP solve (analytic ground truth), laid out as the
correspondence selection (row A8) hands them over: image_points [B,HW,2] px, model_points [B,HW,3] m, counts [B]."""
import numpy as np

from tests.ransac_cases import rand_rot

LM_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])


def make_pnp_case(B=4, HW=4096, n=1500, noise_px=1.0, outliers=0.3, seed=0, K=LM_K):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 777])))
    ip = np.zeros((B, HW, 2), np.float32)
    mp = np.zeros((B, HW, 3), np.float32)
    counts = np.zeros(B, np.int32)
    Rs, ts = np.zeros((B, 3, 3)), np.zeros((B, 3))
    clean = np.zeros((B, HW), bool)
    ratios = np.broadcast_to(np.asarray(outliers, dtype=np.float64), (B,))
    ns = np.broadcast_to(np.asarray(n), (B,))
    for b in range(B):
        nb = int(ns[b])
        ext = rng.random(3) * 0.2 + 0.05
        pts = (rng.random((nb, 3)) - 0.5) * ext
        R, t = rand_rot(rng), np.array([rng.random() * 0.2 - 0.1, rng.random() * 0.2 - 0.1, rng.random() * 0.8 + 0.5])
        Xc = pts @ R.T + t
        uv = (Xc[:, :2] / Xc[:, 2:3]) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
        uv += rng.standard_normal((nb, 2)) * noise_px
        bad = rng.random(nb) < ratios[b]
        uv[bad] += (rng.random((int(bad.sum()), 2)) - 0.5) * 200 + 20
        ip[b, :nb], mp[b, :nb] = uv.astype(np.float32), pts.astype(np.float32)
        counts[b] = nb
        Rs[b], ts[b] = R, t
        clean[b, :nb] = ~bad
    cams = np.broadcast_to(K.astype(np.float32).reshape(1, 9), (B, 9)).copy()
    return dict(image_points=ip, model_points=mp, counts=counts, cams=cams, R=Rs, t=ts, clean=clean, B=B, HW=HW)
