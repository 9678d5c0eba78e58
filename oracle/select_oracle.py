"""ORACLE - TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline may import this; the product never does).

numpy restatement of row A8, the correspondence selection in front of the PnP solve (paths relative to /root/reference):

  core/gdrn_modeling/engine_utils.py:102-115   get_out_coor   (one channel per axis: cat)
  core/gdrn_modeling/engine_utils.py:118-136   get_out_mask   (MASK_LOSS_TYPE L1: per-sample (m - min) / (max - min), no epsilon;
                                               BCE: sigmoid; CE: arg-max over two channels)
  core/gdrn_modeling/gdrn_evaluator.py:89-126  get_img_model_points_with_coords2d

Parity is PINNED: tests/golden/select_golden.npz holds the outputs of those three reference functions on the seeded cases of
tests/select_cases.py (tools/oracle/gen_select_golden.py, run under this container's numpy 2.2: `0.0001 * extent[c]` is an
fp32 product there - python floats are weak scalars since NEP 50); tests/test_select_oracle.py checks this file bit for bit.
`max_num_points` (a random.shuffle subsample, off at the reference's call site) is not restated.
"""
import numpy as np


def out_mask_l1(mask):
    """(B,1,H,W) fp32 -> per-sample min-max normalised mask (engine_utils.py:125-129); 0/0 -> NaN like the reference"""
    m = np.asarray(mask, dtype=np.float32)
    B = m.shape[0]
    mx = m.reshape(B, -1).max(axis=1).reshape(B, 1, 1, 1)
    mn = m.reshape(B, -1).min(axis=1).reshape(B, 1, 1, 1)
    with np.errstate(all="ignore"):
        return ((m - mn) / (mx - mn)).astype(np.float32)


def out_mask(mask, mask_loss_type="L1"):
    """get_out_mask (engine_utils.py:118-136) for cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE L1 | BCE | CE -> (B,1,H,W) fp32.
    BCE = torch.sigmoid in fp32 (torch's own kernel: the golden file holds its output; 1 / (1 + exp(-x)) here is within two ulps);
    CE = arg-max over the two channels as 0.0 / 1.0 (the reference returns int64; the evaluator compares it with mask_thr).
    Pinned by tests/golden/mask_types_golden.npz (tools/oracle/gen_mask_types_golden.py)."""
    if mask_loss_type == "L1":
        return out_mask_l1(mask)
    m = np.asarray(mask, dtype=np.float32)
    if mask_loss_type == "BCE":
        assert m.shape[1] == 1, m.shape
        with np.errstate(over="ignore"):
            return (np.float32(1.0) / (np.float32(1.0) + np.exp(-m))).astype(np.float32)
    if mask_loss_type == "CE":
        return np.argmax(m, axis=1, keepdims=True).astype(np.float32)  # first maximum on a tie, like torch.argmax
    raise NotImplementedError(f"unknown mask loss type: {mask_loss_type}")


def select_correspondences(mask_norm_hw, xyz_hwc, coord2d_hw2, im_H, im_W, extent, mask_thr=0.5):
    """one crop: (H,W) normalised mask, (H,W,3) coordinates in [0,1], (H,W,2) 2D coordinates in [0,1], extent (3,) ->
    image_points (n,2) fp32 pixels, model_points (n,3) fp32 metres, selection mask (H,W) bool; row-major pixel order"""
    e = np.asarray(extent, dtype=np.float32)
    xyz = np.asarray(xyz_hwc, dtype=np.float32).copy()
    c2 = np.asarray(coord2d_hw2, dtype=np.float32).copy()
    for c in range(3):
        xyz[:, :, c] = (xyz[:, :, c] - np.float32(0.5)) * e[c]          # :106-108
    c2[:, :, 0] = c2[:, :, 0] * np.float32(im_W)                         # :110-111
    c2[:, :, 1] = c2[:, :, 1] * np.float32(im_H)
    with np.errstate(invalid="ignore"):
        sel = np.asarray(mask_norm_hw, dtype=np.float32) > np.float32(mask_thr)   # :113-118
    for c in range(3):
        sel = sel & (np.abs(xyz[:, :, c]) > np.float32(0.0001) * e[c])
    return c2[sel].reshape(-1, 2), xyz[sel].reshape(-1, 3), sel
