"""ORACLE - TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's ROI crop construction
(core/gdrn_modeling/data_loader.py:478-627, core/utils/data_utils.py:81-152) INCLUDING the third-party
cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) arithmetic it relies on.

PARITY UNPINNED: cv2 is opencv-python==4.5.5.62 (requirements.txt:118), absent from /root/reference and not installed
here.  warp_affine_bilinear() restates OpenCV 4.5.5 modules/imgproc/src/imgwarp.cpp (cv::warpAffine ->
WarpAffineInvoker: inverse map in double, AB_BITS=10 fixed point, round_delta = AB_SCALE/INTER_TAB_SIZE/2 = 16,
INTER_BITS=5; remapBilinear: 15-bit integer weights + (1<<14) >> 15 for uint8, float weights for float32).
The HIP kernel (csrc/crop_builder.hip) is held to THIS file."""
import numpy as np


def forward_affine(center, scale, out):
    """get_affine_transform(center, scale, 0, out) in closed form: u' = (out/scale)(u - c) + out/2"""
    s = float(out) / float(scale)
    return np.array([[s, 0.0, out * 0.5 - s * float(center[0])], [0.0, s, out * 0.5 - s * float(center[1])]], dtype=np.float64)


def invert_affine(M):
    """cv2.warpAffine's in-place inversion (imgwarp.cpp: D = M0*M4 - M1*M3 ...)"""
    m = M.reshape(-1).astype(np.float64).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _coords(minv, out):
    xs = np.arange(out)
    adelta = np.rint(minv[0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(minv[3] * xs * 1024).astype(np.int64)
    X0 = np.rint((minv[1] * xs + minv[2]) * 1024).astype(np.int64) + 16   # indexed by y
    Y0 = np.rint((minv[4] * xs + minv[5]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    return X >> 5, Y >> 5, X & 31, Y & 31


def warp_affine_bilinear(img, M, out):
    """cv2.warpAffine(img, M, (out,out), flags=INTER_LINEAR) for uint8 or float32 images (H,W[,C])"""
    minv = invert_affine(M)
    sx, sy, ax, ay = _coords(minv, out)
    H, W = img.shape[:2]
    im = img.reshape(H, W, -1)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, 0)

    v00, v01, v10, v11 = tap(sy, sx), tap(sy, sx + 1), tap(sy + 1, sx), tap(sy + 1, sx + 1)
    if img.dtype == np.uint8:
        w00, w01, w10, w11 = (32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32
        acc = (v00.astype(np.int64) * w00[..., None] + v01.astype(np.int64) * w01[..., None] + v10.astype(np.int64) * w10[..., None] +
               v11.astype(np.int64) * w11[..., None] + (1 << 14)) >> 15
        res = acc.astype(np.uint8)
    else:
        fx, fy = ax.astype(np.float32) * np.float32(1 / 32), ay.astype(np.float32) * np.float32(1 / 32)
        one = np.float32(1)
        f00, f01, f10, f11 = (one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx
        res = v00.astype(np.float32) * f00[..., None]
        res = res + v01.astype(np.float32) * f01[..., None]
        res = res + v10.astype(np.float32) * f10[..., None]
        res = res + v11.astype(np.float32) * f11[..., None]
    return res.reshape((out, out) + img.shape[2:])


def build_roi(image_u8, depth, K, center, scale, R=256, out_res=64):
    """one ROI exactly as data_loader.py:523-627 builds it -> roi_img (6,R,R) f32, roi_coord_2d (5,R/4,R/4) f32, resize_ratio"""
    H, W = depth.shape
    roi_img = warp_affine_bilinear(image_u8, forward_affine(center, scale, R), R).transpose(2, 0, 1)
    roi_img = (roi_img - np.zeros((3, 1, 1))) / np.full((3, 1, 1), 255.0)        # normalize_image with mean 0 / std 255
    resize_ratio = out_res / scale
    depth2 = warp_affine_bilinear(depth.astype(np.float32), forward_affine(center, scale, R), R)[:, :, None]
    ymap, xmap = np.mgrid[0:R, 0:R].astype(np.float32)
    Hm = forward_affine(center, scale, R)
    off = np.zeros((3, 3)); off[:2, :] = Hm; off[2, 2] = 1
    depth2 = depth2 / resize_ratio
    newK = np.matmul(off, K)
    pt2 = depth2.astype(np.float32)
    pt0 = (xmap[:, :, None] - newK[0][2]) * pt2 / newK[0][0]
    pt1 = (ymap[:, :, None] - newK[1][2]) * pt2 / newK[1][1]
    depth_xyz = np.concatenate((pt0, pt1, pt2), axis=2).transpose(2, 0, 1)
    roi_img = np.concatenate((roi_img, depth_xyz), axis=0).astype("float32")
    x = np.linspace(0, 1, W, dtype=np.float32); y = np.linspace(0, 1, H, dtype=np.float32)
    coord_2d = np.asarray(np.meshgrid(x, y)).transpose(1, 2, 0)
    rc = warp_affine_bilinear(coord_2d, forward_affine(center, scale, out_res), out_res).transpose(2, 0, 1)
    roi_coord_2d = np.concatenate((depth_xyz[:, ::4, ::4], rc)).astype("float32")
    return roi_img, roi_coord_2d, resize_ratio, newK
