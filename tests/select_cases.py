"""seeded head outputs for the correspondence-selection row A8 (gdrn_evaluator.py:89-126, engine_utils.py:102-136), shared
by tools/oracle/gen_select_golden.py (real reference) and the tests"""
import numpy as np

IM_H, IM_W = 480, 640


def select_case(seed, B=5, side=64):
    """dict of fp32 arrays: mask (B,1,s,s), coor_x/y/z (B,1,s,s), coord2d (B,2,s,s) in [0,1], extent (B,3).
    Built to hit every branch: blobs of foreground, coordinates exactly at / one ulp around the |xyz| > 1e-4*extent filter,
    mask values exactly at the threshold after normalisation, a crop with fewer than 4 survivors, a constant mask
    ((m-min)/(max-min) = 0/0 = NaN: nothing selected)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 4711])))
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
    mask = np.empty((B, 1, side, side), np.float32)
    for b in range(B):
        cx, cy, r = rng.random(3) * np.array([side / 2, side / 2, side / 4]) + np.array([side / 4, side / 4, side / 8])
        blob = ((xx - cx) ** 2 + (yy - cy) ** 2 < r * r).astype(np.float32)
        mask[b, 0] = blob * (0.6 + 0.4 * rng.random((side, side), dtype=np.float32)) + 0.1 * rng.standard_normal((side, side)).astype(np.float32)
    coor = rng.random((B, 3, side, side), dtype=np.float32)
    ext = (rng.random((B, 3), dtype=np.float32) * np.float32(0.2) + np.float32(0.05)).astype(np.float32)
    # coordinates at the centre value and a few ulps around the filter boundary |c - 0.5| * e  vs  1e-4 * e
    coor[:, 0, 10:20, 10:40] = np.float32(0.5)
    for k, d in enumerate((1e-4, 1.0000001e-4, 0.9999999e-4, -1e-4, 2e-4, 5e-5)):
        coor[:, 1, 22 + k, 5:60] = np.float32(0.5) + np.float32(d)
    coor[:, 2, 30:33, :] = np.nextafter(np.float32(0.5001), np.float32(1.0))
    coord2d = rng.random((B, 2, side, side), dtype=np.float32)
    if B >= 4:
        mask[B - 2, 0] = np.float32(0.0)
        mask[B - 2, 0, 3, 3:6] = np.float32(1.0)      # three survivors only (n < 4: the -100 sentinel branch)
        mask[B - 2, 0, 7, 7] = np.float32(0.5)        # exactly AT the threshold: not selected (strict >)
        mask[B - 1, 0] = np.float32(0.25)             # constant mask: 0/0
    return {"mask": mask, "coor_x": coor[:, 0:1].copy(), "coor_y": coor[:, 1:2].copy(), "coor_z": coor[:, 2:3].copy(),
            "coord2d": coord2d, "extent": ext}


def mask_logits_case(mask, mask_loss_type, seed):
    """the mask of select_case() as LOGITS for get_out_mask's BCE / CE branches: (B,1,s,s) for "BCE" (sigmoid; both signs occur, a
    few values exactly 0 = sigmoid exactly 0.5 = not selected at thr 0.5), (B,2,s,s) for "CE" (arg-max; a block of exact ties =
    channel 0 wins = not selected)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 815])))
    m = np.asarray(mask, dtype=np.float32)
    z = ((m - np.float32(0.45)) * np.float32(6.0)).astype(np.float32)
    z[:, 0, 40, 10:20] = np.float32(0.0)
    if mask_loss_type == "BCE":
        return z
    other = (np.float32(0.5) * rng.standard_normal(m.shape)).astype(np.float32)
    two = np.concatenate([other, other + z], axis=1).astype(np.float32)  # channel 1 - channel 0 = z (up to rounding)
    two[:, 1, 41, 10:20] = two[:, 0, 41, 10:20]  # exact ties
    return two
