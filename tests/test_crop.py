"""GPU crop builder vs its numpy oracle (cv2.warpAffine arithmetic restated; parity with cv2 itself unpinned)."""
import numpy as np
import pytest
import torch

from oracle import crop_oracle


def _frames(seed, N=2, H=480, W=640):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 31])))
    img = rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (0.8 + 0.1 * np.sin(xx / 37.0) + 0.05 * np.cos(yy / 23.0))[None].repeat(N, 0).astype(np.float32)
    depth[rng.random((N, H, W)) < 0.05] = 0
    return img, depth


def test_oracle_warp_is_identity_and_shift():
    img, depth = _frames(0, 1, 64, 64)
    M = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(crop_oracle.warp_affine_bilinear(img[0], M, 64), img[0])
    M = np.array([[1.0, 0, 3.0], [0, 1.0, -2.0]])   # dst(x,y) = src(x-3, y+2)
    w = crop_oracle.warp_affine_bilinear(depth[0], M, 64)
    assert np.array_equal(w[:60, 5:], depth[0][2:62, 2:61]) and (w[:, :3] == 0).all()
    # half-pixel shift of a ramp interpolates exactly
    ramp = np.tile(np.arange(64, dtype=np.float32), (64, 1))
    w = crop_oracle.warp_affine_bilinear(ramp, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), 64)
    assert np.allclose(w[:, 1:], ramp[:, 1:] - 0.5)


@pytest.mark.gpu
def test_hip_crop_builder_matches_oracle():
    from rdpn6d_amd.crop import build_crops
    from rdpn6d_amd.synth import LM_K

    dev = torch.device("cuda:0")
    img, depth = _frames(3)
    boxes = np.array([[200.0, 150.0, 330.0, 260.0], [10.0, 20.0, 90.0, 180.0], [500.0, 300.0, 639.0, 470.0], [300.5, 100.25, 380.75, 231.0]])
    idx = np.array([0, 1, 1, 0])
    cams = np.stack([LM_K.astype(np.float32)] * 4)
    out = build_crops(torch.from_numpy(img).to(dev), torch.from_numpy(depth).to(dev), idx, boxes, cams)
    torch.cuda.synchronize()
    for i in range(4):
        c = np.array([0.5 * (boxes[i, 0] + boxes[i, 2]), 0.5 * (boxes[i, 1] + boxes[i, 3])])
        scale = min(max(boxes[i, 2] - boxes[i, 0], boxes[i, 3] - boxes[i, 1], 1) * 1.5, 640) * 1.0
        roi_img, roi_c2d, ratio, _ = crop_oracle.build_roi(img[idx[i]], depth[idx[i]], cams[i], c, scale)
        got_img, got_c2d = out["roi_img"][i].cpu().numpy(), out["roi_coord_2d"][i].cpu().numpy()
        assert np.array_equal(got_img[:3], roi_img[:3]), i                       # uint8 fixed-point path: bit exact
        assert np.abs(got_img[3:] - roi_img[3:]).max() <= 1e-6 * max(1.0, np.abs(roi_img[3:]).max()), i
        assert np.abs(got_c2d - roi_c2d).max() <= 1e-6 * max(1.0, np.abs(roi_c2d).max()), i
        assert abs(out["resize_ratio"][i].item() - ratio) < 1e-6
