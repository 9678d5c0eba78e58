"""ORACLE - TEST INFRASTRUCTURE ONLY.  numpy restatements of

  * region / residual training targets: core/utils/data_utils.py:229-244 (xyz_to_region: cdist + argmin, delta) and
    core/gdrn_modeling/data_loader.py:883-903 (rotate by the GT pose, normalise by the extent, float32 cast)
  * pose errors: lib/pysixd/pose_error.py:297-312 (add), :315-337 (adi, exact nearest neighbour = what cKDTree returns),
    :400-415 (re), :425-436 (te); lib/pysixd/misc.py:895-905 (transform_pts_Rt)

Parity is PINNED: tests/golden/targets_eval_golden.npz was produced by calling the reference's own functions
(tools/oracle/gen_targets_eval_golden.py); tests/test_targets_eval.py checks these restatements against it."""
import numpy as np


def region_targets(xyz_crop, fps_points, R, extent):
    """xyz_crop (H,W,3) f32, fps_points (K,3) f64, R (3,3), extent (3,) -> roi_xyz (3,H,W) f32, roi_region (H,W) int32"""
    bh, bw = xyz_crop.shape[:2]
    mask = ((xyz_crop[:, :, 0] != 0) | (xyz_crop[:, :, 1] != 0) | (xyz_crop[:, :, 2] != 0)).astype("uint8")
    x = xyz_crop.reshape(bh * bw, 3).astype(np.float64)
    d = np.sqrt(((x[:, None, :] - fps_points[None].astype(np.float64)) ** 2).sum(-1))
    ids = np.argmin(d, axis=1).reshape(bh, bw) + 1
    delta = xyz_crop - fps_points[ids - 1]
    delta = R.dot(delta.reshape(-1, 3).T).T.reshape((bh, bw, 3))
    roi = delta.transpose(2, 0, 1).copy()
    for c in range(3):
        roi[c] = roi[c] / extent[c] + 0.5
    return roi.astype("float32"), (mask * ids).astype(np.int32)


def transform_pts_Rt(pts, R, t):
    return (R.dot(pts.T) + t.reshape((3, 1))).T


def add(R_est, t_est, R_gt, t_gt, pts):
    return np.linalg.norm(transform_pts_Rt(pts, R_est, t_est) - transform_pts_Rt(pts, R_gt, t_gt), axis=1).mean()


def adi(R_est, t_est, R_gt, t_gt, pts):
    pe, pg = transform_pts_Rt(pts, R_est, t_est), transform_pts_Rt(pts, R_gt, t_gt)
    out = 0.0
    for i in range(0, len(pg), 512):  # exact NN by brute force, chunked
        d = ((pg[i:i + 512, None, :] - pe[None]) ** 2).sum(-1)
        out += np.sqrt(d.min(1)).sum()
    return out / len(pg)


def re(R_est, R_gt):
    tr = np.trace(np.dot(R_est, R_gt.T))
    tr = tr if tr <= 3 else 3
    return np.rad2deg(np.arccos(min(1.0, max(-1.0, 0.5 * (tr - 1.0)))))


def te(t_est, t_gt):
    return np.linalg.norm(t_gt.flatten() - t_est.flatten())
