"""Seeded synthetic inputs and weights for the RDPN6D hot path.

Everything here is generated with ``numpy.random.Generator(PCG64)`` so the GPU box can
regenerate, bit for bit, what the oracle / the reference saw in the build container
(SURVEY.md §8d).  The SHA-256 of the blobs is committed under ``tests/golden/`` and checked.

Shapes and distributions follow the reference's loader contract:
  * ``roi_img``  (B,6,R,R)   = RGB/255 + depth back-projected with the crop-adjusted K
                               (/root/reference/core/gdrn_modeling/data_loader.py:523-576,605)
  * ``roi_coord_2d`` (B,5,R/4,R/4) = depth_xyz[:, ::4, ::4] + warped 2D coordinate grid
                               (data_loader.py:618-627)
  * crop affine: centre -> (R/2, R/2), zoom R/scale
                               (/root/reference/core/utils/data_utils.py:108-152)
  * ``scale = 1.5*max(w,h)``, ``resize_ratio = out_res/scale`` (data_loader.py:478-488)
"""
import hashlib
import zlib

import numpy as np

LM_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=np.float64)
YCBV_K = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]], dtype=np.float64)
IM_W, IM_H = 640, 480


def _rng(seed, *salt):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed)] + [int(s) for s in salt])))


def _smooth_noise(rng, B, R):
    """Low-frequency field in [-1,1]: 9x9 uniform grid, bilinearly upsampled to RxR."""
    g = rng.random((B, 9, 9), dtype=np.float32) * 2.0 - 1.0
    t = np.linspace(0.0, 8.0, R, dtype=np.float32)
    i0 = np.minimum(np.floor(t).astype(np.int64), 7)
    f = (t - i0).astype(np.float32)
    rows = g[:, i0, :] * (1 - f)[None, :, None] + g[:, i0 + 1, :] * f[None, :, None]  # (B,R,9)
    out = rows[:, :, i0] * (1 - f)[None, None, :] + rows[:, :, i0 + 1] * f[None, None, :]
    return out.astype(np.float32)


def ellipsoid_points(extent, n, seed):
    """n seeded points on the surface of an axis-aligned ellipsoid with the given extent (3,)."""
    rng = _rng(seed, 77)
    v = rng.standard_normal((n, 3), dtype=np.float32)
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return (v * (0.5 * np.asarray(extent, dtype=np.float32))[None, :]).astype(np.float32)


def make_inputs(B, seed=0, res=256, num_regions=32, cam="lm"):
    """Synthetic batch in the reference's ``batch_data`` contract (numpy, fp32 / int64)."""
    R = int(res)
    out_res = R // 4
    K = LM_K if cam == "lm" else YCBV_K
    rng = _rng(seed, 1)
    rgb = rng.random((B, 3, R, R), dtype=np.float32)
    z0 = rng.random(B, dtype=np.float32) * np.float32(1.1) + np.float32(0.4)
    depth = z0[:, None, None] + np.float32(0.05) * _smooth_noise(rng, B, R)
    holes = rng.random((B, R, R), dtype=np.float32) < np.float32(0.1)
    depth = np.where(holes, np.float32(0.0), depth).astype(np.float32)

    centers = np.stack(
        [rng.random(B, dtype=np.float32) * 320 + 160, rng.random(B, dtype=np.float32) * 240 + 120], axis=1
    ).astype(np.float32)
    whs = (rng.random((B, 2), dtype=np.float32) * 160 + 40).astype(np.float32)
    scale = np.minimum(np.float32(1.5) * whs.max(axis=1), np.float32(max(IM_H, IM_W))).astype(np.float32)
    resize_ratio = (np.float32(out_res) / scale).astype(np.float32)
    extents = (rng.random((B, 3), dtype=np.float32) * np.float32(0.2) + np.float32(0.05)).astype(np.float32)

    # crop affine u' = s*(u - cx) + R/2 ; K' = A @ K  (data_loader.py:553-566)
    s = (np.float64(R) / scale.astype(np.float64))
    A = np.zeros((B, 3, 3), dtype=np.float64)
    A[:, 0, 0] = s
    A[:, 1, 1] = s
    A[:, 0, 2] = R / 2.0 - s * centers[:, 0].astype(np.float64)
    A[:, 1, 2] = R / 2.0 - s * centers[:, 1].astype(np.float64)
    A[:, 2, 2] = 1.0
    Kp = A @ K[None]
    xs = np.arange(R, dtype=np.float32)
    pt2 = (depth / resize_ratio[:, None, None]).astype(np.float32)
    pt0 = ((xs[None, None, :] - Kp[:, 0, 2][:, None, None]) * pt2 / Kp[:, 0, 0][:, None, None]).astype(np.float32)
    pt1 = ((xs[None, :, None] - Kp[:, 1, 2][:, None, None]) * pt2 / Kp[:, 1, 1][:, None, None]).astype(np.float32)
    depth_xyz = np.stack([pt0, pt1, pt2], axis=1).astype(np.float32)
    x = np.concatenate([rgb, depth_xyz], axis=1).astype(np.float32)

    # 2D coordinate grid (linspace(0,1) over the full image) warped into the out_res crop; zero outside
    so = np.float64(out_res) / scale.astype(np.float64)
    xo = np.arange(out_res, dtype=np.float64)
    u_img = (xo[None, :] - out_res / 2.0) / so[:, None] + centers[:, 0:1].astype(np.float64)  # (B,out_res)
    v_img = (xo[None, :] - out_res / 2.0) / so[:, None] + centers[:, 1:2].astype(np.float64)
    cu = np.where((u_img >= 0) & (u_img <= IM_W - 1), u_img / (IM_W - 1), 0.0)
    cv = np.where((v_img >= 0) & (v_img <= IM_H - 1), v_img / (IM_H - 1), 0.0)
    coord2d = np.stack(
        [np.broadcast_to(cu[:, None, :], (B, out_res, out_res)), np.broadcast_to(cv[:, :, None], (B, out_res, out_res))],
        axis=1,
    ).astype(np.float32)
    roi_coord_2d = np.concatenate([depth_xyz[:, :, ::4, ::4], coord2d], axis=1).astype(np.float32)

    fps = np.zeros((B, num_regions, 3), dtype=np.float32)
    for b in range(B):
        # region anchors: the first K points of a seeded ellipsoid-surface sample of the object
        # (the real pipeline picks them with the fps kernel offline; any K distinct surface
        # points exercise the same arithmetic)
        fps[b] = ellipsoid_points(extents[b], num_regions, seed * 1000 + b)

    ncls = 13 if cam == "lm" else 21
    return {
        "roi_img": x,
        "roi_coord_2d": roi_coord_2d,
        "fps": fps,
        "roi_cam": np.broadcast_to(K.astype(np.float32)[None], (B, 3, 3)).copy(),
        "roi_center": centers,
        "roi_wh": whs,
        "resize_ratio": resize_ratio,
        "roi_extent": extents,
        "roi_cls": (np.arange(B) % ncls).astype(np.int64),
    }


def make_train_gt(B, inputs, seed=5):
    """Seeded GT tensors in the batch_data contract (engine_utils.py:6-63)."""
    rng = _rng(seed, 99)
    r = int(inputs["roi_coord_2d"].shape[-1])  # output resolution (64 for 256x256 crops)
    sc = np.float32(r / 64.0)                  # blob geometry scales with it (exactly 1 at the reference resolution)
    yy, xx = np.mgrid[0:r, 0:r].astype(np.float32)
    gt = {}
    cx = (rng.random(B, dtype=np.float32) * 20 + 22) * sc
    cy = (rng.random(B, dtype=np.float32) * 20 + 22) * sc
    rad = (rng.random(B, dtype=np.float32) * 10 + 12) * sc
    blob = (((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) < rad[:, None, None] ** 2)
    gt["roi_mask_visib"] = blob.astype(np.float32)
    gt["roi_mask_trunc"] = blob.astype(np.float32)
    gt["roi_mask_obj"] = blob.astype(np.float32)
    gt["roi_region"] = (rng.integers(1, 33, size=(B, r, r)) * blob).astype(np.int64)
    gt["roi_xyz"] = (rng.random((B, 3, r, r), dtype=np.float32) * blob[:, None]).astype(np.float32)
    q = rng.standard_normal((B, 4)).astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                  1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                  1 - 2 * (x * x + y * y)], axis=1).reshape(B, 3, 3)
    gt["ego_rot"] = R.astype(np.float32)
    gt["trans"] = np.stack([rng.random(B) * 0.3 - 0.15, rng.random(B) * 0.3 - 0.15, rng.random(B) + 0.4], 1).astype(np.float32)
    gt["roi_trans_ratio"] = (rng.standard_normal((B, 3)) * 0.3).astype(np.float32)
    gt["roi_points"] = ((rng.random((B, 3000, 3), dtype=np.float32) - 0.5) * inputs["roi_extent"][:, None, :]).astype(np.float32)
    return gt


def make_state_dict(shapes, seed=1234):
    """Seeded O(1)-activation weights for an ordered {name: shape} mapping (numpy fp32).

    Per-tensor generator keyed by crc32(name) so the result is independent of key order.
    conv / FC weight ~ N(0, 2/fan_in); norm gamma ~ U[0.5,1.5]; every bias ~ 0.1*N(0,1);
    running_mean = 0, running_var = 1, num_batches_tracked = 0 (BN statistics are then
    *calibrated* and shipped as a fixture - see tests/golden/README.md)."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        rng = _rng(seed, zlib.crc32(name.encode()))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = np.zeros(shape, dtype=np.float32)
        elif leaf == "running_var":
            out[name] = np.ones(shape, dtype=np.float32)
        elif leaf == "weight" and len(shape) == 1:
            out[name] = (rng.random(shape, dtype=np.float32) + np.float32(0.5)).astype(np.float32)
        elif leaf == "weight":
            if len(shape) == 4 and name.endswith("rot_head_net.features.0.weight"):
                fan_in = shape[0] * shape[2] * shape[3] / 4.0  # ConvTranspose (Cin,Cout,k,k), stride 2
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
            else:
                fan_in = shape[-1]
            std = np.float32(np.sqrt(2.0 / fan_in))
            out[name] = (rng.standard_normal(shape, dtype=np.float32) * std).astype(np.float32)
        elif leaf == "bias":
            out[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)).astype(np.float32)
        else:
            raise KeyError(f"unexpected state_dict entry {name}")
    return out


# The second, WELL-CONDITIONED fixture (tests/golden/model_c1w.npz): the same seeded recipe with the last BatchNorm of every
# residual branch damped (gamma x 0.1), as a trained ResNet has it (zero-gamma initialisation of Goyal et al.; trained
# residual branches are small corrections to the identity).  The plain random-weight net of make_state_dict() amplifies
# fp32 round-off ~100x from stem to head, which puts the REFERENCE's own fp32 output 5e-4 away from the exact result;
# with this recipe the reference is 5e-5 (maps) / 2e-5 (pose) from exact and the north star's bare 1e-4 can be asserted.
# The input seed of that fixture is chosen so that no pixel's region arg-max is a tie at the tolerance (the reference's
# smallest top-2 logit gap is 1.1e-4; with seed 0 three pixels sit below 1e-4 and the reference itself flips them
# between 1 and 8 threads).
C1W_INPUT_SEED = 36
# the TRAINING pass of that fixture runs with batch statistics, i.e. on different maps: its batch is seeded separately, by the
# same criterion (smallest top-2 gap of the train-mode forward 1.3e-4), because one flipped arg-max pixel changes three
# ConvPnPNet input channels and with them every ConvPnPNet gradient by ~1e-3
C1W_TRAIN_INPUT_SEED = 50
C1W_RESIDUAL_GAMMA = 0.1


def make_trained_like_state_dict(shapes, seed=1234, residual_gamma=C1W_RESIDUAL_GAMMA):
    sd = make_state_dict(shapes, seed=seed)
    for k in sd:
        if k.startswith("backbone.layer") and k.endswith(("bn2.weight", "bn3.weight")):
            blk_has_bn3 = k.rsplit(".", 2)[0] + ".bn3.weight" in sd
            if k.endswith("bn3.weight") or not blk_has_bn3:  # the LAST norm of the branch (bn2 for BasicBlock, bn3 for Bottleneck)
                sd[k] = (sd[k] * np.float32(residual_gamma)).astype(np.float32)
    return sd


def sha256_of(arrays):
    """SHA-256 over the raw bytes of a list of C-contiguous arrays (order matters)."""
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
