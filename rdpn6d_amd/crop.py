"""Host side of the GPU crop builder: derives the per-ROI scalars the way the reference's
loader does (core/gdrn_modeling/data_loader.py:478-566: scale = max(bw,bh)*DZI_PAD_SCALE clipped to the image,
resize_ratio = out_res/scale, K' = A @ K) and launches rdpn6d_crop_builder_f32 on full frames that already sit in HBM."""
import ctypes

import numpy as np
import torch

from . import _lib


def _fwd(center, scale, out):
    s = float(out) / float(scale)
    return np.array([[s, 0.0, out * 0.5 - s * float(center[0])], [0.0, s, out * 0.5 - s * float(center[1])]], dtype=np.float64)


def _inv(M):
    m = M.reshape(-1).astype(np.float64).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _fwd_batch(centers, scale, out):
    """_fwd for all crops: [B, 6] rows (s, 0, tx, 0, s, ty) - the same operations per element as the scalar form"""
    s = float(out) / scale
    z = np.zeros_like(s)
    return np.stack([s, z, out * 0.5 - s * centers[:, 0], z, s, out * 0.5 - s * centers[:, 1]], 1)


def _inv_batch(M6):
    """_inv for all crops, operation by operation as the scalar form (cv2.invertAffineTransform's order): [B, 6] -> [B, 6]"""
    m = M6.astype(np.float64).copy()
    D = m[:, 0] * m[:, 4] - m[:, 1] * m[:, 3]
    with np.errstate(divide="ignore"):
        D = np.where(D != 0, 1.0 / D, 0.0)
    A11, A22 = m[:, 4] * D, m[:, 0] * D
    m[:, 0] = A11
    m[:, 1] *= -D
    m[:, 3] *= -D
    m[:, 4] = A22
    b1 = -m[:, 0] * m[:, 2] - m[:, 1] * m[:, 5]
    b2 = -m[:, 3] * m[:, 2] - m[:, 4] * m[:, 5]
    m[:, 2], m[:, 5] = b1, b2
    return m


def build_crops(images_u8, depths, img_idx, bboxes_xyxy, cams, input_res=256, out_res=64, pad_scale=1.5):
    """images_u8 [N,H,W,3] uint8 and depths [N,H,W] float32 on the GPU; img_idx [B]; bboxes [B,4] xyxy (host); cams [B,3,3] (host).
    Returns the reference's per-ROI tensors: roi_img [B,6,R,R], roi_coord_2d [B,5,R/4,R/4] (device) and
    bbox_center [B,2], scale [B], roi_wh [B,2], resize_ratio [B] (device float32).  The per-ROI scalars are derived for the whole
    batch at once and go up in three copies (fp64 block, fp32 block, frame indices)."""
    if not images_u8.is_cuda:
        raise RuntimeError("rdpn6d_amd.crop: frames must live on the GPU (no CPU fallback)")
    N, H, W, _ = images_u8.shape
    bb = np.asarray(bboxes_xyxy, dtype=np.float64)
    B = bb.shape[0]
    centers = np.stack([0.5 * (bb[:, 0] + bb[:, 2]), 0.5 * (bb[:, 1] + bb[:, 3])], 1)
    bw, bh = np.maximum(bb[:, 2] - bb[:, 0], 1), np.maximum(bb[:, 3] - bb[:, 1], 1)
    scale = np.minimum(np.maximum(bh, bw) * pad_scale, max(H, W)) * 1.0
    ratio = out_res / scale
    fwd_in = _fwd_batch(centers, scale, input_res)
    inv_in = _inv_batch(fwd_in)
    inv_out = _inv_batch(_fwd_batch(centers, scale, out_res))
    off = np.zeros((B, 3, 3))
    off[:, :2] = fwd_in.reshape(B, 2, 3)
    off[:, 2, 2] = 1
    nk = np.matmul(off, np.asarray(cams, dtype=np.float64).reshape(B, 3, 3))
    Kn = np.stack([nk[:, 0, 0], nk[:, 1, 1], nk[:, 0, 2], nk[:, 1, 2]], 1)
    dev = images_u8.device
    d64 = torch.from_numpy(np.concatenate([inv_in.ravel(), inv_out.ravel(), Kn.ravel(), ratio.ravel()])).to(dev)  # 17 B doubles
    d_inv_in, d_inv_out, d_K, d_ratio = d64[: 6 * B], d64[6 * B: 12 * B], d64[12 * B: 16 * B], d64[16 * B:]
    d_idx = torch.from_numpy(np.ascontiguousarray(np.asarray(img_idx), dtype=np.int32)).to(dev)
    f32 = torch.from_numpy(np.concatenate([centers.ravel(), scale, bw, bh, ratio]).astype(np.float32)).to(dev)
    roi_img = torch.empty(B, 6, input_res, input_res, dtype=torch.float32, device=dev)
    roi_c2d = torch.empty(B, 5, input_res // 4, input_res // 4, dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(_lib.load().rdpn6d_crop_builder_f32(P(images_u8.contiguous()), P(depths.float().contiguous()), N, H, W, P(d_idx), P(d_inv_in),
                                                   P(d_inv_out), P(d_K), P(d_ratio), B, input_res, P(roi_img), P(roi_c2d),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "crop_builder")
    return {"roi_img": roi_img, "roi_coord_2d": roi_c2d, "bbox_center": f32[: 2 * B].view(B, 2), "scale": f32[2 * B: 3 * B],
            "roi_wh": torch.stack([f32[3 * B: 4 * B], f32[4 * B: 5 * B]], 1), "resize_ratio": f32[5 * B:]}
