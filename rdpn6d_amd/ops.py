"""Functional wrappers over the C ABI (torch tensors in, torch tensors out; HIP kernels do the work).

Used by the tests and by callers that want one operator rather than the whole GDRN plan.
Every function raises if its tensors are not on the GPU: there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .gdrn import _pad_to, _pad_vec, _ptr, pack_conv_weight


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("rdpn6d_amd.ops: tensors must live on the GPU (no CPU fallback)")


def conv2d_nhwc(x, weight, scale=None, shift=None, stride=1, pad=0, residual=None, act=0, slope=0.0, out=None,
                out_co=0, in_co=0, cin=None, out_f32=False):
    """x NHWC [B,H,W,Cs] fp32 (Cs multiple of 4), weight OIHW (torch layout).  Returns NHWC [B,Ho,Wo,N]
    (or writes channels [out_co, out_co+N) of ``out``).  A bfloat16 ``x`` selects the bf16 matrix-pipe kernel
    (Cs multiple of 8, reduction width padded to 32; output and residual bf16, or both fp32 with ``out_f32``)."""
    _need_gpu(x, weight, scale, shift, residual, out)
    lib = _lib.load()
    B, H, W, cs = x.shape
    N, wcin, k, _ = weight.shape
    cin_real = cin or wcin
    bf = x.dtype == torch.bfloat16
    cin_pad = _pad_to(cin_real, 32 if bf else 16)
    assert in_co + cin_pad <= cs, "input slice must cover the padded reduction width"
    wp = pack_conv_weight(weight.float(), cin_pad=cin_pad)
    if bf:
        wp = wp.to(torch.bfloat16)
        # the residual has the output's dtype: bf16, or fp32 together with an fp32 output
        assert residual is None or residual.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, N, dtype=torch.bfloat16 if bf and not out_f32 else torch.float32, device=x.device)
    sc = _pad_vec(scale.float(), wp.shape[0], 1.0) if scale is not None else None
    sh = _pad_vec(shift.float(), wp.shape[0], 0.0) if shift is not None else None
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(x), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(residual), _ptr(out)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, W, cin_pad, cs, in_co
    d.Ho, d.Wo, d.stride = Ho, Wo, stride
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps):
        d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = N, wp.shape[0], Ho, Wo
    d.osy = d.osx = 1
    d.ooy = d.oox = 0
    d.out_cs, d.out_co = out.shape[-1], out_co
    if residual is not None:
        d.res_cs, d.res_co = residual.shape[-1], 0
    d.act, d.slope = act, slope
    if bf:
        assert out.dtype == (torch.float32 if out_f32 else torch.bfloat16)
        _lib.check(lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 1 if out_f32 else 0, _stream()), "conv2d_bf16")
    else:
        _lib.check(lib.rdpn6d_conv2d_f32(ctypes.byref(d), _stream()), "conv2d")
    return out


def split_bf16x3(x):
    """fp32 tensor -> [3, *x.shape] bf16 planes with x == p0 + p1 + p2 to 2^-27 (input format of conv2d_nhwc_x3)."""
    _need_gpu(x)
    x = x.contiguous()
    n = x.numel()
    pe = _pad_to(n, 8)
    planes = torch.empty(3, pe, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().rdpn6d_split_bf16x3(_ptr(x), n, _ptr(planes), pe, _stream()), "split_bf16x3")
    return planes


def conv2d_nhwc_x3(x, weight, scale=None, shift=None, stride=1, pad=0, residual=None, act=0, slope=0.0, want_planes=False,
                   residual_planes=None):
    """fp32-accurate convolution on the bf16 matrix pipe (rdpn6d_conv2d_bf16x3): x NHWC fp32 [B,H,W,C] or its planes from
    split_bf16x3 / a previous call (tuple (planes, shape)); weight OIHW fp32.  Returns y fp32 NHWC and, with want_planes,
    (planes, shape) of y for the next layer."""
    lib = _lib.load()
    if isinstance(x, tuple):
        xp, (B, H, W, cs) = x
    else:
        B, H, W, cs = x.shape
        xp = split_bf16x3(x)
    _need_gpu(xp, weight, scale, shift, residual)
    N, wcin, k, _ = weight.shape
    cin_pad = _pad_to(wcin, 16)
    assert cin_pad <= cs
    wp32 = pack_conv_weight(weight.float(), cin_pad=cin_pad)
    wp = split_bf16x3(wp32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32, device=xp.device)
    npe = _pad_to(out.numel(), 8)
    yp = torch.empty(3, npe, dtype=torch.bfloat16, device=xp.device) if want_planes else None
    sc = _pad_vec(scale.float(), wp32.shape[0], 1.0) if scale is not None else None
    sh = _pad_vec(shift.float(), wp32.shape[0], 0.0) if shift is not None else None
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(xp), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(residual), _ptr(out)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, W, cin_pad, cs, 0
    d.Ho, d.Wo, d.stride = Ho, Wo, stride
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps):
        d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = N, wp32.shape[0], Ho, Wo
    d.osy = d.osx = 1
    d.ooy = d.oox = 0
    d.out_cs, d.out_co = N, 0
    if residual is not None:
        d.res_cs, d.res_co = residual.shape[-1], 0
    d.act, d.slope = act, slope
    rp = None
    if residual_planes is not None:  # (planes, shape) of an NHWC tensor with the output's geometry
        rp = residual_planes[0]
        d.res_cs, d.res_co = residual_planes[1][-1], 0
    if not lib.rdpn6d_conv_bf16x3_eligible(ctypes.byref(d)):
        raise ValueError("layer not eligible for the bf16x3 kernels (Cin % 16, N % 8, aligned slices)")
    _lib.check(lib.rdpn6d_conv2d_bf16x3_ex(ctypes.byref(d), xp.shape[1], wp.shape[1], _ptr(yp), npe, _ptr(rp),
                                           rp.shape[1] if rp is not None else 0, _stream()), "conv2d_bf16x3")
    return (out, (yp, tuple(out.shape))) if want_planes else out


def split_h2(x):
    """fp32 NHWC tensor [..., C] (C % 32 == 0) -> its two-plane fp16 form [pixels][C/32][2][32] holding 16*x = hi + lo (input
    format of conv2d_nhwc_h2); also returns the device overflow flag (1 when a value beyond +-4094 had to be clamped)."""
    _need_gpu(x)
    x = x.contiguous()
    C = x.shape[-1]
    npix = x.numel() // C
    h2 = torch.empty(npix, C // 32, 2, 32, dtype=torch.float16, device=x.device)
    flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().rdpn6d_split_h2(_ptr(x), C, 0, C, _ptr(h2), npix, _ptr(flag), _stream()), "split_h2")
    return h2, flag


def merge_h2(h2, shape):
    """inverse of split_h2 (exact): [pixels][C/32][2][32] fp16 -> fp32 tensor of `shape`"""
    return ((h2[:, :, 0].float() + h2[:, :, 1].float()) / 16.0).reshape(shape)


def conv2d_nhwc_h2(x, weight, scale=None, shift=None, stride=1, pad=0, residual=None, act=0, slope=0.0, want_h2=False, residual_h2=None,
                   split_k=True, wfrag=False):
    """fp32-accurate convolution on the fp16 matrix pipe, two planes per operand (rdpn6d_conv2d_h2): x NHWC fp32 [B,H,W,C] or
    (h2 tensor, shape) from split_h2 / a previous call; weight OIHW fp32 (Cin % 32 == 0).  Returns y fp32 NHWC and, with
    want_h2, ((h2 tensor, shape) of y, overflow flag).  split_k=False withholds the workspace, so a launch that would cut K into
    slices (rdpn6d_conv_h2_workspace_bytes != 0: too few tiles to fill the chip) runs un-split.  wfrag: rdpn6d_conv2d_h2_wf."""
    from .gdrn import pack_h2_weight

    lib = _lib.load()
    if isinstance(x, tuple):
        xh, (B, H, W, cs) = x
    else:
        B, H, W, cs = x.shape
        xh, _ = split_h2(x)
    _need_gpu(xh, weight, scale, shift, residual)
    N, wcin, k, _ = weight.shape
    assert wcin % 32 == 0 and wcin <= cs
    wp32 = pack_conv_weight(weight.float(), cin_pad=wcin)
    wh, inv = pack_h2_weight(wp32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float32, device=xh.device)
    yh = torch.empty(B * Ho * Wo, N // 32, 2, 32, dtype=torch.float16, device=xh.device) if want_h2 else None
    flag = torch.zeros(1, dtype=torch.int32, device=xh.device)
    sc = _pad_vec(scale.float(), wp32.shape[0], 1.0) * inv if scale is not None else inv
    sh = _pad_vec(shift.float(), wp32.shape[0], 0.0) if shift is not None else None
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(xh), _ptr(wh), _ptr(sc), _ptr(sh), _ptr(residual), _ptr(out)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, W, wcin, cs, 0
    d.Ho, d.Wo, d.stride = Ho, Wo, stride
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps):
        d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = N, wp32.shape[0], Ho, Wo
    d.osy = d.osx = 1
    d.ooy = d.oox = 0
    d.out_cs, d.out_co = N, 0
    if residual is not None:
        d.res_cs, d.res_co = residual.shape[-1], 0
    d.act, d.slope = act, slope
    rh = None
    if residual_h2 is not None:
        rh = residual_h2[0]
        d.res_cs, d.res_co = residual_h2[1][-1], 0
    if not lib.rdpn6d_conv_h2_kernel_for(ctypes.byref(d)):
        raise ValueError("layer not eligible for the h2 kernels (Cin % 32, N % 8, aligned slices)")
    ws_bytes = int(lib.rdpn6d_conv_h2_workspace_bytes(ctypes.byref(d))) if split_k else 0
    if wfrag:  # the kernel form that loads its weight fragments from L2 (fragment-major weights), where the layer's kernel has one
        if not lib.rdpn6d_conv_h2_wfrag_wanted(ctypes.byref(d)):
            raise ValueError("this layer's kernel has no weights-from-L2 form")
        wf = torch.empty_like(wh)
        _lib.check(lib.rdpn6d_h2_weight_frag(_ptr(wh), wh.shape[0], d.ntaps, wcin // 32, _ptr(wf), _stream()), "h2_weight_frag")
        _lib.check(lib.rdpn6d_conv2d_h2_wf(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), _ptr(wf), _stream()), "conv2d_h2_wf")
    elif ws_bytes:  # a launch too small to fill the chip: K slices + a fixed-order reduce
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xh.device)
        _lib.check(lib.rdpn6d_conv2d_h2_ws(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), None, _ptr(ws), ws_bytes, _stream()), "conv2d_h2")
    else:
        _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), _stream()), "conv2d_h2")
    return (out, ((yh, tuple(out.shape)), flag)) if want_h2 else out


def conv2d_h2_colmax(x, weight, scale, shift, rows_per_group):
    """The column-max form of a 1x1 / kxk h2 convolution (rdpn6d_conv2d_h2_colmax + rdpn6d_h2_colmax_decode): x NHWC fp32 [B,H,W,C], weight
    OIHW -> the h2 record [groups][N/32][2][32] fp16 of max over each group's rows of scale * conv + shift (stride 1, 'same' padding).
    The activation itself is never written."""
    from .gdrn import pack_h2_weight

    lib = _lib.load()
    B, H, W, cs = x.shape
    xh, _ = split_h2(x)
    N, wcin, k, _ = weight.shape
    wp32 = pack_conv_weight(weight.float(), cin_pad=wcin)
    wh, inv = pack_h2_weight(wp32)
    sc = _pad_vec(scale.float(), wp32.shape[0], 1.0) * inv
    sh = _pad_vec(shift.float(), wp32.shape[0], 0.0)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.res, d.y = _ptr(xh), _ptr(wh), _ptr(sc), _ptr(sh), None, None
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, W, wcin, cs, 0
    d.Ho, d.Wo, d.stride = H, W, 1
    taps = [(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps):
        d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = N, wp32.shape[0], H, W
    d.osy = d.osx = 1
    d.ooy = d.oox = 0
    d.out_cs, d.out_co, d.act, d.slope = N, 0, 0, 0.0
    if not lib.rdpn6d_conv_h2_colmax_ok(ctypes.byref(d), rows_per_group):
        raise ValueError("layer not eligible for the column-max form (256x256 kernel, rows_per_group % 256 == 0)")
    groups = B * H * W // rows_per_group
    keys = torch.zeros(groups, d.Npad, dtype=torch.int64, device=x.device)
    out = torch.empty(groups, N // 32, 2, 32, dtype=torch.float16, device=x.device)
    flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(lib.rdpn6d_conv2d_h2_colmax(ctypes.byref(d), _ptr(keys), rows_per_group, _ptr(flag), _stream()), "conv2d_h2_colmax")
    _lib.check(lib.rdpn6d_h2_colmax_decode(_ptr(keys), groups, N, d.Npad, _ptr(out), _stream()), "h2_colmax_decode")
    return out, keys, flag


def stem_conv7x7(x_nchw, weight, scale, shift):
    _need_gpu(x_nchw, weight, scale, shift)
    B, xc, R, _ = x_nchw.shape
    w = weight.float().permute(0, 2, 3, 1).contiguous()
    y = torch.empty(B, R // 2, R // 2, 64, dtype=torch.float32, device=x_nchw.device)
    _lib.check(_lib.load().rdpn6d_stem_conv7x7_f32(_ptr(x_nchw), B, xc, R, _ptr(w), _ptr(scale), _ptr(shift), _ptr(y),
                                                   _stream()), "stem")
    return y


def maxpool3x3s2(x):
    _need_gpu(x)
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rdpn6d_maxpool3x3s2_f32(_ptr(x), B, H, W, C, _ptr(y), _stream()), "maxpool")
    return y


def upsample_bilinear(x, factor):
    _need_gpu(x)
    B, H, W, C = x.shape
    y = torch.empty(B, H * factor, W * factor, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rdpn6d_upsample_bilinear_f32(_ptr(x), B, H, W, C, factor, _ptr(y), _stream()), "upsample")
    return y


def global_max_concat_(buf, C):
    _need_gpu(buf)
    B, H, W, cs = buf.shape
    _lib.check(_lib.load().rdpn6d_global_max_concat_f32(_ptr(buf), B, H * W, C, cs, _stream()), "global_max_concat")
    return buf


def groupnorm_relu_(x, groups, gamma, beta):
    _need_gpu(x, gamma, beta)
    B, H, W, C = x.shape
    _lib.check(_lib.load().rdpn6d_groupnorm_relu_f32(_ptr(x), B, H * W, C, groups, _ptr(gamma), _ptr(beta), _stream()), "gn")
    return x


def farthest_point_sampling(pts, sn, init_center=False, start=None, return_index=False):
    """Host-array face of the fps ABI (core/csrc/fps/fps_utils.py:6-21): returns pts[idxs] (sn,3) f32, exactly what the reference's
    function returns (get_fps_and_center, core/utils/data_utils.py:217-226, concatenates it with the centroid row).  Extras, both off by
    default: ``start`` pins the random variant's start index, ``return_index=True`` returns (pts[idxs], idxs) for the parity tests."""
    pts = np.ascontiguousarray(pts, np.float32)
    pn, three = pts.shape
    assert three == 3
    idxs = np.zeros([sn], np.int32)
    lib = _lib.load()
    P = ctypes.c_void_p
    if init_center:
        _lib.check(lib.rdpn6d_fps_host(pts.ctypes.data_as(P), idxs.ctypes.data_as(P), pn, sn, -1), "fps")
    elif start is not None:
        _lib.check(lib.rdpn6d_fps_host(pts.ctypes.data_as(P), idxs.ctypes.data_as(P), pn, sn, int(start)), "fps")
    else:
        lib.farthest_point_sampling(pts.ctypes.data_as(P), idxs.ctypes.data_as(P), pn, sn)
        if (idxs < 0).any():
            raise RuntimeError("farthest_point_sampling failed: " + lib.rdpn6d_last_error().decode())
    return (pts[idxs], idxs) if return_index else pts[idxs]


def ransac_kabsch(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax, mask_thr=0.5, inlier_thr=0.01,
                  iters=100, confidence=0.99, seed=0, net_pose=None, net_mode="ransac", max_t_diff=1.0, split=True):
    """Per-crop RANSAC + Kabsch on the dense maps (device tensors).  Returns pose [B,12] (R row-major | t),
    n_inliers [B] int32, inlier_mask [B,HW] uint8, best_hyp [B] int32.  Role of process_pnp_ransac
    (gdrn_evaluator.py:316-435) for the RGB-D residual formulation; sentinel pose -100 when < 3 points."""
    _need_gpu(out_nchw, coord2d, fps, extents, resize_ratios, region_argmax)
    B, C = out_nchw.shape[0], out_nchw.shape[1]
    HW = out_nchw[0, 0].numel()
    K = C - 5
    dev = out_nchw.device
    pose = torch.empty(B, 12, dtype=torch.float32, device=dev)
    nin = torch.empty(B, dtype=torch.int32, device=dev)
    mask = torch.empty(B, HW, dtype=torch.uint8, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    args = [t.contiguous() for t in (out_nchw, coord2d, fps, extents, resize_ratios, region_argmax)]
    assert args[5].dtype == torch.int32
    lib = _lib.load()
    # (workspace of the split form: with fewer crops than CUs a crop's hypotheses are spread over up to four workgroups - same results)
    ws = torch.empty(int(lib.rdpn6d_ransac_workspace_bytes(B)) if split else 1, dtype=torch.uint8, device=dev)
    # process_net_and_pnp (gdrn_evaluator.py:187-314): the learned pose [B,12] initialises / guards the solve
    npz = net_pose.float().contiguous() if net_pose is not None else None
    _lib.check(lib.rdpn6d_ransac_kabsch_ws(*[_ptr(t) for t in args], _ptr(npz), B, HW, K, mask_thr, inlier_thr, iters, confidence, seed,
                                           {"ransac": 1, "iter": 2}[net_mode], max_t_diff, _ptr(pose), _ptr(nin), _ptr(mask), _ptr(best),
                                           _ptr(ws) if split else None, ws.numel() if split else 0, _stream()), "ransac_kabsch")
    return pose, nin, mask, best


def ransac_pnp(image_points, model_points, counts, cams, reproj_thr=3.0, iters=100, confidence=0.99, seed=0, net_pose=None,
               net_mode="ransac", max_t_diff=1.0, minimal="p3p", split=True):
    """2D-3D RANSAC-PnP per crop (rdpn6d_ransac_pnp_f32; the role of misc.pnp_v2 -> cv2.solvePnPRansac(EPnP, 3 px, 100 iterations) at
    gdrn_evaluator.py:316-435).  image_points [B,HW,2] px, model_points [B,HW,3] m, counts [B] int32 from
    select_correspondences; cams [B,3,3].  With net_pose [B,12]: net_mode "ransac" = the learned pose is hypothesis 0, "iter" =
    Gauss-Newton from it over all correspondences (process_net_and_pnp).  Returns pose [B,12] (R row-major | t; -100 when fewer
    than 4 correspondences), n_inliers [B], inlier_mask [B,HW] (indexed like the lists), best_hyp [B].
    minimal (cfg.TEST.PNP_MINIMAL): "p3p" = P3P + 1 on sets of four with a Gauss-Newton refit (default) | "epnp" = the solver the
    reference's call names (flags=cv2.SOLVEPNP_EPNP): EPnP on sets of five, EPnP refit on the inliers."""
    _need_gpu(image_points, model_points, counts, cams)
    B, HW = image_points.shape[0], image_points.shape[1]
    dev = image_points.device
    pose = torch.empty(B, 12, dtype=torch.float32, device=dev)
    nin = torch.empty(B, dtype=torch.int32, device=dev)
    mask = torch.empty(B, HW, dtype=torch.uint8, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    args = [image_points.float().contiguous(), model_points.float().contiguous(), counts.contiguous(), cams.float().reshape(B, 9).contiguous()]
    assert args[2].dtype == torch.int32
    npz = net_pose.float().contiguous() if net_pose is not None else None
    mode = 0 if net_pose is None else {"ransac": 1, "iter": 2}[net_mode]
    lib = _lib.load()
    # (workspace of the split form: with fewer crops than CUs a crop's hypotheses are spread over several workgroups - same results)
    ws = torch.empty(int(lib.rdpn6d_ransac_pnp_workspace_bytes(B)) if split else 1, dtype=torch.uint8, device=dev)
    _lib.check(lib.rdpn6d_ransac_pnp_ws(*[_ptr(t) for t in args], _ptr(npz), B, HW, float(reproj_thr), int(iters), float(confidence),
                                        int(seed), mode, float(max_t_diff), {"p3p": 0, "epnp": 1}[str(minimal).lower()], _ptr(pose),
                                        _ptr(nin), _ptr(mask), _ptr(best), _ptr(ws) if split else None, ws.numel() if split else 0, _stream()),
               "ransac_pnp")
    return pose, nin, mask, best


def region_targets(xyz_hwc, fps64, rot, extent):
    """Training targets on device (data_utils.xyz_to_region + data_loader.py:881-903): xyz_hwc [B,H,W,3] f32 model-space
    crop, fps64 [B,K,3] float64 anchors, rot [B,3,3], extent [B,3] -> roi_xyz [B,3,H,W] f32, roi_region [B,H,W] int64."""
    _need_gpu(xyz_hwc, fps64, rot, extent)
    B, H, W, _ = xyz_hwc.shape
    K = fps64.shape[1]
    assert fps64.dtype == torch.float64
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=xyz_hwc.device)
    reg = torch.empty(B, H, W, dtype=torch.int64, device=xyz_hwc.device)
    args = [xyz_hwc.float().contiguous(), fps64.contiguous(), rot.float().contiguous(), extent.float().contiguous()]
    _lib.check(_lib.load().rdpn6d_region_targets_f32(*[_ptr(t) for t in args], B, H * W, K, _ptr(out), _ptr(reg), _stream()), "region_targets")
    return out, reg


def pose_errors(R_est, t_est, R_gt, t_gt, pts):
    """ADD / ADI / re[deg] / te per pose on device in float64 (lib/pysixd/pose_error.py add, adi, re, te).
    R_* [B,3,3], t_* [B,3], pts [n,3] or [B,n,3] -> [B,4] float64"""
    _need_gpu(R_est, t_est, R_gt, t_gt, pts)
    B = R_est.shape[0]
    est = torch.cat([R_est.reshape(B, 9), t_est.reshape(B, 3)], 1).double().contiguous()
    gt = torch.cat([R_gt.reshape(B, 9), t_gt.reshape(B, 3)], 1).double().contiguous()
    p = pts.double().contiguous()
    per_pose = 1 if p.dim() == 3 else 0
    n = p.shape[-2]
    out = torch.empty(B, 4, dtype=torch.float64, device=est.device)
    scratch = torch.empty(B, n, 3, dtype=torch.float64, device=est.device) if n * 24 > 150 * 1024 else None
    _lib.check(_lib.load().rdpn6d_pose_errors_f64(_ptr(est), _ptr(gt), _ptr(p), per_pose, n, B, _ptr(scratch), _ptr(out), _stream()), "pose_errors")
    return out


def select_correspondences(out_maps, coord2d, extents, im_H, im_W, mask_thr=0.5, u_ch=None, v_ch=None, return_masks=False,
                           mask_loss_type="L1"):
    """Row A8 on device (engine_utils.get_out_coor / get_out_mask + gdrn_evaluator.get_img_model_points_with_coords2d):
    out_maps [B,C,H,W] (channel 0 mask, 1..3 coor_x/y/z), coord2d [B,C2,H,W] (u, v in the LAST two channels unless u_ch / v_ch
    say otherwise), extents [B,3] -> image_points [B,HW,2], model_points [B,HW,3], counts [B] int32 (first counts[b] rows valid;
    row-major pixel order); with return_masks also the selection mask [B,H,W] uint8 and the normalised mask [B,H,W].
    mask_loss_type = cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE: "L1" per-crop min-max | "BCE" sigmoid | "CE" arg-max over TWO mask
    channels (out_maps channels 0, 1 = mask, 2..4 = coor_x/y/z) - get_out_mask's three branches."""
    from .gdrn import MASK_TYPES

    _need_gpu(out_maps, coord2d, extents)
    B, C, H, W = out_maps.shape
    C2 = coord2d.shape[1]
    u_ch = C2 - 2 if u_ch is None else u_ch
    v_ch = C2 - 1 if v_ch is None else v_ch
    dev = out_maps.device
    ip = torch.empty(B, H * W, 2, dtype=torch.float32, device=dev)
    mp = torch.empty(B, H * W, 3, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    sel = torch.empty(B, H, W, dtype=torch.uint8, device=dev) if return_masks else None
    nm = torch.empty(B, H, W, dtype=torch.float32, device=dev) if return_masks else None
    args = [out_maps.float().contiguous(), coord2d.float().contiguous(), extents.float().contiguous()]
    _lib.check(_lib.load().rdpn6d_select_correspondences_mt_f32(_ptr(args[0]), C, _ptr(args[1]), C2, u_ch, v_ch, _ptr(args[2]), None, int(im_H),
                                                                int(im_W), B, H * W, float(mask_thr), MASK_TYPES[mask_loss_type], _ptr(ip),
                                                                _ptr(mp), _ptr(cnt), _ptr(sel), _ptr(nm), _stream()), "select_correspondences")
    return (ip, mp, cnt, sel, nm) if return_masks else (ip, mp, cnt)
