"""Synthetic dense maps with a KNOWN pose for the RANSAC/Kabsch solve (analytic ground truth).

Per foreground pixel: region a, model point p = anchor[a] + d, camera point P = R p + t (+ noise),
residual delta = R d (+ noise)  =>  P - delta = R anchor[a] + t  (SURVEY.md section 0).
The maps are laid out exactly as the HIP path produces / consumes them:
out_nchw [B,4+K+1,HW] (mask | residual/extent+0.5 | region logits), coord2d [B,5,HW] (P / ratio | uv)."""
import numpy as np


def rand_rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_case(B=3, K=32, side=64, noise=0.001, outliers=0.3, holes=0.1, seed=0, fg_radius=22):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 4242])))
    HW = side * side
    C = 4 + K + 1
    out = np.zeros((B, C, HW), np.float32)
    cd = np.zeros((B, 5, HW), np.float32)
    fps = np.zeros((B, K, 3), np.float32)
    ext = (rng.random((B, 3)) * 0.2 + 0.05).astype(np.float32)
    ratio = (rng.random(B) * 0.4 + 0.25).astype(np.float32)
    am = rng.integers(0, K, size=(B, HW)).astype(np.int32)
    Rs, ts, clean = np.zeros((B, 3, 3)), np.zeros((B, 3)), np.zeros((B, HW), bool)
    yy, xx = np.mgrid[0:side, 0:side]
    for b in range(B):
        v = rng.standard_normal((K, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        fps[b] = (v * 0.5 * ext[b]).astype(np.float32)
        R, t = rand_rot(rng), np.array([rng.random() * 0.2 - 0.1, rng.random() * 0.2 - 0.1, rng.random() * 0.8 + 0.5])
        Rs[b], ts[b] = R, t
        fg = ((xx - side / 2) ** 2 + (yy - side / 2) ** 2 < fg_radius ** 2).reshape(-1)
        d = (rng.random((HW, 3)) - 0.5) * 0.04
        P = (fps[b][am[b]].astype(np.float64) + d) @ R.T + t + rng.standard_normal((HW, 3)) * noise
        delta = d @ R.T + rng.standard_normal((HW, 3)) * noise
        bad = rng.random(HW) < (outliers[b] if np.ndim(outliers) else outliers)  # per-crop ratio when an array is given
        P[bad] += (rng.random((int(bad.sum()), 3)) - 0.5) * 0.3 + 0.05
        hole = rng.random(HW) < holes
        P[hole] = 0.0
        out[b, 0] = (fg * 1.0 + rng.standard_normal(HW) * 0.05).astype(np.float32)
        out[b, 0, 0], out[b, 0, 1] = -0.2, 1.2  # pin the min / max used by the normalisation
        out[b, 1:4] = (delta / ext[b] + 0.5).T.astype(np.float32)
        out[b, 4:] = rng.standard_normal((K + 1, HW)).astype(np.float32)
        cd[b, :3] = (P / ratio[b]).T.astype(np.float32)
        cd[b, 3:] = rng.random((2, HW)).astype(np.float32)
        clean[b] = fg & ~bad & ~hole
    return dict(out_nchw=out, coord2d=cd, fps=fps, extents=ext, ratios=ratio, argmax=am, R=Rs, t=ts, clean=clean,
                B=B, K=K, HW=HW)


def pose_errors(pose, R, t):
    """rotation error (deg) and translation error (m) of a [12] pose against ground truth."""
    Rp, tp = pose[:9].reshape(3, 3).astype(np.float64), pose[9:].astype(np.float64)
    c = np.clip((np.trace(Rp.T @ R) - 1) / 2, -1, 1)
    return np.degrees(np.arccos(c)), np.linalg.norm(tp - t)
