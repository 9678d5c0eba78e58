"""The two-plane fp16 ("h2") convolution kernels - rdpn6d_conv2d_h2, csrc/conv_igemm_h2.hip - are fp32 convolutions: error
against an fp64 convolution no larger than the fp32-MFMA kernel's own (typical error AND the a-priori bound under
cancellation), exact hand-over of activations between layers, loud range handling."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H2_CASES = [
    # B, H, Cin, Cout, k, stride, res, act
    (2, 32, 256, 256, 3, 1, True, 1),     # head layer shape (tile kernel at this size)
    (64, 16, 256, 256, 3, 1, False, 1),   # 64 row tiles x 1 column tile ... M = 16384 rows: tile kernel
    (3, 30, 64, 256, 3, 1, False, 2),     # ragged M (2700 rows), odd size, leaky
    (2, 32, 32, 512, 1, 1, True, 0),      # 1x1, a single K-chunk
    (2, 32, 128, 256, 3, 2, False, 1),    # stride 2
    (1, 16, 1024, 256, 3, 1, False, 0),   # K = 9216
    (8, 32, 64, 64, 3, 1, True, 1),       # trunk layer1 shape: N = 64
    (4, 32, 128, 128, 3, 1, True, 1),     # layer2 shape
    (2, 16, 64, 128, 3, 2, False, 1),     # stride-2 entry conv of a stage
    (2, 16, 64, 128, 1, 2, False, 0),     # 1x1 stride-2 downsample
    (3, 7, 512, 512, 3, 1, True, 1),      # layer4 shape: odd size, ragged rows, K = 4608
    (12, 64, 256, 256, 3, 1, True, 1),    # 192 tiles of 256 rows: the 256x256 eight-phase kernel
    (11, 64, 64, 256, 3, 1, False, 2),    # eight-phase kernel, 18 K-tiles, ragged last tile (45056 rows = 176 tiles)
    (1, 8, 512, 512, 3, 1, True, 1),      # layer4 of ONE crop: 64 rows, 144 chunks -> tile kernel with K cut into 16 slices
    (1, 15, 256, 256, 3, 1, False, 2),    # split-K with ragged rows (225), no residual, LeakyReLU
    (1, 16, 128, 256, 3, 2, False, 1),    # stride-2 entry convolution of one crop, split-K
    # ---- the 8-wave ping-pong kernel (conv_igemm_h2_pp.hip: N % 128 == 0 and >= 224 tiles) and its neighbours
    (64, 16, 256, 256, 3, 1, True, 1),    # layer3 at B = 64: 256 tiles of 128x128, four LDS stages, h2 residual
    (64, 32, 128, 128, 3, 1, True, 1),    # layer2 at B = 64: 256 tiles of 256x128
    (37, 32, 128, 128, 3, 1, True, 2),    # 296 tiles of 128x128, two rounds, the second one ragged
    (29, 62, 64, 64, 3, 1, True, 1),      # layer1-like, odd size, ragged last tile (111 476 rows): tile kernel (N = 64)
    (64, 8, 512, 512, 3, 1, True, 1),     # layer4 at B = 64: 128 tiles of 128x128 would not fill the chip -> tile kernel
    (28, 16, 256, 256, 3, 1, False, 2),   # 112 x 2 = 224 tiles of 128x128: the smallest launch the ping-pong kernel takes
    (64, 32, 128, 256, 3, 2, False, 1),   # stride-2 entry convolution of layer3 at B = 64
    (64, 32, 256, 128, 1, 1, False, 0),   # 1x1: 8 chunks, the shortest loop the kernel takes
    (40, 16, 256, 384, 3, 1, True, 1),    # three column tiles (N = 384): 80 x 3 = 240 tiles of 128x128
    (64, 32, 256, 512, 1, 2, False, 0),   # Bottleneck-style 1x1 stride-2 downsample: 64 x 4 = 256 tiles of 256x128, 8 chunks
    (30, 32, 320, 128, 1, 1, False, 1),   # 1x1 over 320 channels: 240 tiles of 128x128, 10 chunks (10 % 3 = 1: the ring's remainder step)
    (30, 32, 352, 128, 1, 1, True, 1),    # ... 11 chunks (two remainder steps)
]


def _ref64(x, w, sc, sh, res, k, stride, act):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    y = y * sc.double() + sh.double()
    if res is not None:
        y = y + res.double()
    if act == 1:
        y = y.clamp(min=0)
    elif act == 2:
        y = torch.where(y > 0, y, y * 0.1)
    return y


@pytest.mark.parametrize("case", H2_CASES)
def test_conv_h2_has_fp32_accuracy(case):
    from rdpn6d_amd import _lib, ops
    import ctypes

    dev = torch.device("cuda:0")
    B, H, Cin, Cout, k, stride, use_res, act = case
    g = torch.Generator().manual_seed(sum(case) * 7 + 1)
    x = torch.randn(B, H, H, Cin, generator=g)
    x[0, 0, 0, :8] = torch.tensor([1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 3e-8, 2000.0, -4000.0])  # small magnitudes (lo subnormal) and the top of the range
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g) if use_res else None
    y64 = _ref64(x, w, sc, sh, res, k, stride, act)
    kw = dict(stride=stride, pad=k // 2, act=act, slope=0.1)
    xd, wd, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    resd = res.to(dev) if use_res else None
    y32 = ops.conv2d_nhwc(xd, wd, scd, shd, residual=resd, **kw)
    yh, ((h2, shape), flag) = ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual=resd, want_h2=Cout % 32 == 0, **kw) if Cout % 32 == 0 else (
        ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual=resd, **kw), ((None, None), None))
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    if use_res:  # the residual handed over as an h2 tensor (what chained trunk layers do): the same up to its 22-bit form
        yr = ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual_h2=(ops.split_h2(resd)[0], tuple(resd.shape)), **kw)
        torch.cuda.synchronize()
        assert (yr - yh).abs().max().item() <= 2.0 ** -21 * resd.abs().max().item() + 2.0 ** -22 * y64.abs().max().item()  # (+ one ulp of the sum)
    e32, eh = (y32.cpu().double() - y64).abs(), (yh.cpu().double() - y64).abs()
    scale = y64.abs().max().item()
    print(f"{case}: max err vs fp64  fp32-MFMA {e32.max().item():.3e}  h2 {eh.max().item():.3e}  "
          f"(rms {e32.pow(2).mean().sqrt().item():.2e} / {eh.pow(2).mean().sqrt().item():.2e}, |y|max {scale:.2f})")
    assert eh.max().item() <= 1.25 * e32.max().item() + 1e-7 * scale
    assert eh.pow(2).mean().sqrt().item() <= 1.1 * e32.pow(2).mean().sqrt().item() + 1e-8 * scale
    # the h2 record of the output re-assembles the fp32 output to 2^-22, and feeding it to the next layer equals feeding the fp32 tensor
    back = ops.merge_h2(h2, shape)
    assert (back - yh).abs().max().item() <= 2.0 ** -21 * scale
    if k == 3 and stride == 1 and B * H * H <= 20000:
        w2 = (torch.randn(256, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5).to(dev)
        a1 = ops.conv2d_nhwc_h2(yh, w2, pad=1)
        a2 = ops.conv2d_nhwc_h2((h2, shape), w2, pad=1)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2)
    d = _lib.ConvDesc()
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride, d.ntaps, d.N, d.Npad, d.out_cs = B, H, H, Cin, Cin, Ho, Ho, stride, k * k, Cout, (Cout + 63) // 64 * 64, Cout
    which = _lib.load().rdpn6d_conv_h2_kernel_for(ctypes.byref(d))
    assert which == (2 if B in (11, 12) else 1), which  # the two big cases run on the eight-phase kernel
    d.dy[0] = -(k // 2)  # (the tap offsets do not matter for the queries below)
    _lib.load().rdpn6d_conv_h2_set_wfrag(2)  # (the form is off by default: measured slower; here its bit-identity is what is checked)
    wanted = _lib.load().rdpn6d_conv_h2_wfrag_wanted(ctypes.byref(d))
    assert wanted or case != (64, 16, 256, 256, 3, 1, True, 1)  # layer3 at B = 64 is the shape it was written for
    if wanted:
        # round 5: the ping-pong kernel's 128x128 form loads its WEIGHT fragments straight from L2 (fragment-major copy of the weights,
        # three-slot register ring) instead of staging the weight tile through LDS - same k order per accumulator: bit-identical
        yf = ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual=resd, wfrag=True, **kw)
        torch.cuda.synchronize()
        assert torch.equal(yf, yh), (case, (yf - yh).abs().max().item())
        print(f"{case}: weights-from-L2 form bit-identical")
    _lib.load().rdpn6d_conv_h2_set_wfrag(0)
    wsb = _lib.load().rdpn6d_conv_h2_workspace_bytes(ctypes.byref(d))
    assert wsb > 0 if B == 1 else (wsb == 0 if which == 2 else True), (case, wsb)  # one crop's layers cut K into slices ...
    if wsb:  # ... with the same bits on every run (slices are added in slice order), fp32 rounding away from the un-split launch
        again = ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual=resd, **kw)
        whole = ops.conv2d_nhwc_h2(xd, wd, scd, shd, residual=resd, split_k=False, **kw)
        torch.cuda.synchronize()
        assert torch.equal(again, yh)
        assert not torch.equal(whole, yh) and (whole - yh).abs().max().item() <= 4e-6 * scale


def test_conv_h2_error_bound_under_cancellation():
    """fp32-class error BOUND: products that cancel almost completely (operands in +/- pairs plus a tiny signal, magnitudes
    spread over 2^-10 .. 2^9 - the h2 activation range ends at 4094): |err| <= c * 2^-24 * sum|a||b| with c no larger than the
    fp32-MFMA kernel's fmaf chain, on the eight-phase kernel (B = 40: 160 tiles) and on the tile kernel."""
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    for B in (2, 40):
        g = torch.Generator().manual_seed(99 + B)
        H, C, N = 32, 256, 256
        mag = torch.exp2(torch.randint(-10, 10, (B, H, H, C // 2), generator=g).float())
        half = (torch.randn(B, H, H, C // 2, generator=g) * mag).clamp(-4000, 4000)
        x = torch.cat([half, half], dim=-1)
        wh = torch.randn(N, C // 2, 3, 3, generator=g) / 48.0
        w = torch.cat([wh, -wh], dim=1) + torch.randn(N, C, 3, 3, generator=g) * 1e-6
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
        bound = torch.nn.functional.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs(), padding=1).permute(0, 2, 3, 1) * 2.0 ** -24
        yh = ops.conv2d_nhwc_h2(x.to(dev), w.to(dev), pad=1).cpu().double()
        y1 = ops.conv2d_nhwc(x.to(dev), w.to(dev), pad=1).cpu().double()
        rh, r1 = ((yh - y64).abs() / bound).max().item(), ((y1 - y64).abs() / bound).max().item()
        print(f"cancellation test B={B}: |err| / (2^-24 sum|a||b|): h2 {rh:.3f}  fp32-MFMA {r1:.3f}")
        assert rh <= 1.25 * r1 and rh <= 48.0
        assert y64.abs().max().item() < 1e-2 * bound.max().item() * 2 ** 24


def test_split_h2_is_the_exact_two_term_split():
    """csrc/h2_format.h writes the fp32 -> (hi, lo) split instruction by instruction (v_cvt_pk_f16_f32 for hi, v_fma_mixlo/hi_f16 for
    lo = fp16(s - hi) in one instruction, one unsigned compare for the range check): it must be BIT-identical to the plain definition
    hi = fp16(s), lo = fp16(s - fp32(hi)), s = clamp(16 x), on values from fp16-subnormal lo terms to the top of the range, and the
    flag must fire exactly for |16 x| > 65504, inf and NaN."""
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    mags = torch.tensor([1e-9, 1e-7, 3e-6, 1e-4, 0.01, 0.3, 1.0, 17.0, 333.0, 600.0])  # (|randn| < 6: all inside +-4094)
    x = torch.randn(4096, 64, generator=g) * mags[torch.randint(0, 10, (4096, 64), generator=g)]
    x[0, :8] = torch.tensor([0.0, -0.0, 4094.0, -4094.0, 2.0 ** -24, 65504.0 / 16, 1.0 + 2.0 ** -11, -(1.0 + 2.0 ** -12)])
    h2, flag = ops.split_h2(x.to(dev))
    torch.cuda.synchronize()
    s = (x * 16.0).clamp(-65504.0, 65504.0)
    hi = s.half()
    lo = (s - hi.float()).half()
    got = h2.cpu().reshape(4096, 2, 2, 32)  # [pixel][group of 32 channels][hi | lo][32]
    assert torch.equal(got[:, :, 0].reshape(4096, 64).view(torch.int16), hi.view(torch.int16))
    assert torch.equal(got[:, :, 1].reshape(4096, 64).view(torch.int16), lo.view(torch.int16))
    assert int(flag.item()) == 0
    assert (lo != 0).float().mean().item() > 0.5 and ((lo.float().abs() < 6.2e-5) & (lo != 0)).any()  # subnormal lo terms are in the sample
    for bad, want in ((4094.001, 1), (-1e9, 1), (float("inf"), 1), (float("nan"), 1), (4093.9, 0)):
        y = x.clone()
        y[7, 3] = bad
        _, f = ops.split_h2(y.to(dev))
        assert int(f.item()) == want, bad


def test_h2_range_is_guarded_not_silent():
    """an activation beyond the fp16 range of the h2 format (|a| * 16 > 65504) is clamped and REPORTED, never an inf"""
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    x = torch.ones(1, 8, 8, 32, device=dev)
    x[0, 1, 1, 3] = 5000.0
    h2, flag = ops.split_h2(x)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and torch.isfinite(h2.float()).all()
    w = torch.full((64, 32, 1, 1), 100.0, device=dev)
    y, ((yh, _), flag2) = ops.conv2d_nhwc_h2(torch.ones(1, 8, 8, 32, device=dev) * 10, w, want_h2=True)  # outputs 32000 > 4094
    torch.cuda.synchronize()
    assert int(flag2.item()) == 1 and torch.isfinite(yh.float()).all() and abs(y[0, 0, 0, 0].item() - 32000.0) < 1.0
    _, f3 = ops.split_h2(torch.randn(2, 4, 4, 64, device=dev) * 100)
    assert int(f3.item()) == 0


@pytest.mark.parametrize("B,R", [(2, 256), (3, 64), (1, 320), (2, 72)])
def test_fused_stem_pool_h2_vs_fp64(B, R):
    """rdpn6d_stem_pool_h2: conv1 7x7/2 + folded BN + ReLU + MaxPool2d(3,2,1) in one kernel on the fp16 matrix pipe, pooled output as
    an h2 tensor - against an fp64 evaluation, with the fp32 VALU stem + fp32 max-pool kernels as the yardstick."""
    import ctypes

    from rdpn6d_amd import _lib, ops
    from rdpn6d_amd.gdrn import _ptr, pack_stem_h2_weight

    lib, dev = _lib.load(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + R)
    x = torch.rand(B, 6, R, R, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    y64 = torch.nn.functional.conv2d(x[:, :3].double(), w.double(), stride=2, padding=3) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    y64 = torch.nn.functional.max_pool2d(y64.clamp(min=0), 3, 2, 1).permute(0, 2, 3, 1)
    xd = x.to(dev)
    wh, inv = pack_stem_h2_weight(w.to(dev))
    scf = (sc.to(dev) * inv).contiguous()
    Rp = R // 4
    yh = torch.empty(B * Rp * Rp, 2, 2, 32, dtype=torch.float16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(xd), B, 6, R, _ptr(wh), _ptr(scf), _ptr(sh.to(dev)), _ptr(yh), _ptr(flag), None))
    s32 = ops.stem_conv7x7(xd, w.to(dev), sc.to(dev), sh.to(dev))
    p32 = ops.maxpool3x3s2(s32)
    torch.cuda.synchronize()
    mine = ops.merge_h2(yh, (B, Rp, Rp, 64)).cpu().double()
    e_h2, e_32 = (mine - y64).abs().max().item(), (p32.cpu().double() - y64).abs().max().item()
    scale = y64.abs().max().item()
    print(f"B={B} R={R}: fused h2 stem+pool vs fp64 {e_h2:.3e} | fp32 VALU stem + pool vs fp64 {e_32:.3e} (|y|max {scale:.2f})")
    assert int(flag.item()) == 0 and e_h2 <= 1.25 * e_32 + 2.0 ** -21 * scale  # (+ the 22-bit h2 record of the output itself)
    # the same kernel storing bf16 / fp16 NHWC (front of the 16-bit inference mode): one rounding of the same values
    for fmt, dt, ulp in ((1, torch.bfloat16, 2.0 ** -8), (2, torch.float16, 2.0 ** -11)):
        y16 = torch.empty(B, Rp, Rp, 64, dtype=dt, device=dev)
        _lib.check(lib.rdpn6d_stem_pool_h2_ex(_ptr(xd), B, 6, R, _ptr(wh), _ptr(scf), _ptr(sh.to(dev)), _ptr(y16), fmt, None, None))
        torch.cuda.synchronize()
        err = (y16.cpu().double() - y64).abs()
        assert (err <= ulp * y64.abs() * 1.01 + e_h2 + 1e-7).all(), (fmt, err.max().item())  # (unit roundoff 2^-8 | 2^-11)
        assert torch.equal(y16.cpu(), mine.to(dt)) or (y16.cpu().double() - mine).abs().max().item() <= ulp * scale  # = the h2 result rounded once (up to double rounding)
    # round 5: the pooled forms take the 3x3 / stride-2 max in registers (stem_pool_h2_v2_kernel: m-tiles = stem rows, one cross-half
    # exchange per odd pooled column, two small LDS hand-offs) - the same convolution values, an exact max: BIT-IDENTICAL to the
    # round-3 kernel that pooled through LDS (out_fmt + 0x100), in all three output formats
    for fmt, ref_t in ((0, yh), (1, torch.empty(B, Rp, Rp, 64, dtype=torch.bfloat16, device=dev)), (2, torch.empty(B, Rp, Rp, 64, dtype=torch.float16, device=dev))):
        new_t, old_t = torch.full_like(ref_t, 7.0), torch.full_like(ref_t, 9.0)
        _lib.check(lib.rdpn6d_stem_pool_h2_ex(_ptr(xd), B, 6, R, _ptr(wh), _ptr(scf), _ptr(sh.to(dev)), _ptr(new_t), fmt, _ptr(flag), None))
        _lib.check(lib.rdpn6d_stem_pool_h2_ex(_ptr(xd), B, 6, R, _ptr(wh), _ptr(scf), _ptr(sh.to(dev)), _ptr(old_t), fmt | 0x100, _ptr(flag), None))
        torch.cuda.synchronize()
        assert torch.equal(new_t.view(torch.int16), old_t.view(torch.int16)), (fmt, (new_t.float() - old_t.float()).abs().max().item())
    # the RAW stem convolution (training forward, out_fmt 3 | 4: no ReLU, no pooling) with the device-packed weight record
    wh2, inv2 = torch.empty_like(wh), torch.empty_like(inv)
    _lib.check(lib.rdpn6d_stem_pack_h2(_ptr(w.to(dev).contiguous()), _ptr(wh2), _ptr(inv2), None))
    torch.cuda.synchronize()
    assert torch.equal(wh2, wh) and torch.equal(inv2, inv)  # = gdrn.pack_stem_h2_weight
    raw64 = torch.nn.functional.conv2d(x[:, :3].double(), w.double(), stride=2, padding=3).permute(0, 2, 3, 1)
    zero = torch.zeros(64, device=dev)
    for fmt, dt, ulp in ((3, torch.bfloat16, 2.0 ** -8), (4, torch.float16, 2.0 ** -11)):
        yr = torch.full((B, R // 2, R // 2, 64), float("nan"), dtype=dt, device=dev)
        _lib.check(lib.rdpn6d_stem_pool_h2_ex(_ptr(xd), B, 6, R, _ptr(wh2), _ptr(inv2), _ptr(zero), _ptr(yr), fmt, None, None))
        torch.cuda.synchronize()
        err = (yr.cpu().double() - raw64).abs()
        assert torch.isfinite(yr).all() and (err <= ulp * raw64.abs() * 1.01 + 2e-6 * raw64.abs().max().item()).all(), (fmt, err.max().item())


def test_global_max_record_and_constant_input_bias_of_the_conv_transpose():
    """cfg.TEST.FOLD_GLOBAL_MAX building blocks.  rdpn6d_global_max_h2 = the per-crop channel max of an h2 tensor, exactly (as a
    record).  rdpn6d_convt3x3s2_const_bias_f32 = what ConvTranspose2d(3, 2, 1, output_padding 1) makes of a spatially constant
    input, per output parity and border position - against torch's conv_transpose2d in fp64 on a constant map."""
    import ctypes

    from rdpn6d_amd import _lib, ops
    from rdpn6d_amd.gdrn import _ptr

    lib, dev = _lib.load(), torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(5)
    B, HW, C = 3, 200, 128
    x = torch.randn(B, HW, C, generator=g).to(dev)
    x[1, 17] = x[1, 3]  # ties
    xh, _ = ops.split_h2(x)
    out = torch.zeros(B, C // 32, 2, 32, dtype=torch.float16, device=dev)
    _lib.check(lib.rdpn6d_global_max_h2(_ptr(xh), B, HW, C, C, _ptr(out), st))
    torch.cuda.synchronize()
    assert torch.equal(ops.merge_h2(out, (B, C)), ops.merge_h2(xh, (B, HW, C)).max(dim=1).values)

    B, Cc, F, H = 2, 48, 64, 5
    gm = torch.randn(B, Cc, generator=g).double()
    W = torch.randn(Cc, F, 3, 3, generator=g).double() / Cc ** 0.5
    sc = (torch.rand(F, generator=g) + 0.5).double()
    ref = torch.nn.functional.conv_transpose2d(gm[:, :, None, None].expand(B, Cc, H, H), W, stride=2, padding=1, output_padding=1) * sc[None, :, None, None]
    V = torch.einsum("bc,cnyx->byxn", gm, W).reshape(B, 9 * F).float().to(dev).contiguous()
    tab = torch.zeros(4, B, 4, F, device=dev)
    _lib.check(lib.rdpn6d_convt3x3s2_const_bias_f32(_ptr(V), _ptr(sc.float().to(dev)), B, F, _ptr(tab), st))
    torch.cuda.synchronize()
    OH = 2 * H
    checked = 0
    for py in (0, 1):
        for px in (0, 1):
            for lr in (0, 1):
                for lc in (0, 1):
                    if (lr and not py) or (lc and not px):
                        continue  # the last output row / column is odd
                    oy, ox = (OH - 1 if lr else 2 + py), (OH - 1 if lc else 4 + px)
                    got = tab[py * 2 + px, :, lr * 2 + lc].cpu().double()
                    assert (got - ref[:, :, oy, ox]).abs().max().item() < 1e-5, (py, px, lr, lc)
                    checked += 1
    assert checked == 9


@pytest.mark.parametrize("backbone,res", [(34, 256), (50, 320)])
def test_folded_global_max_plan_equals_the_concat_plan(backbone, res):
    """the two forms of the same network: [l3 | broadcast max] through a 1024-channel ConvTranspose, and l3 through a 512-channel one
    plus the per-crop bias of the constant half (cfg.TEST.FOLD_GLOBAL_MAX); and xyz_emb (1x1 convolution + BatchNorm) evaluated on
    layer4's 8x8 map before the bilinear up-sampling instead of after it (cfg.TEST.CONV_BEFORE_UPSAMPLE); and conv3 + BatchNorm
    (no activation) composed into the ConvTranspose weights (cfg.TEST.COMPOSE_CONV3_CONVT) - outputs equal up to fp32 summation
    order (on the ill-conditioned stress weights
    of the bench, where the reference's own fp32 evaluation is 4e-4 from the exact one - DESIGN.md section 2 - the two orders may
    differ by a fraction of that; the well-conditioned c1w parity tests run on the folded plan)"""
    import bench
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    B = 16  # >= 16: the h2 plan.  (50, 320): Bottleneck trunk, 40x40 -> 80x80 ConvTranspose (other border rows, other tile counts)
    t = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(B, seed=11, res=res).items()}
    outs = {}
    for fold in (True, False):
        model, _ = bench.build_model(dev, "none", backbone=backbone, res=res)
        model.cfg.TEST.USE_PNP = False
        model.cfg.TEST.FOLD_GLOBAL_MAX = fold
        model.cfg.TEST.CONV_BEFORE_UPSAMPLE = fold
        model.cfg.TEST.COMPOSE_CONV3_CONVT = fold
        with torch.no_grad():
            o = bench.step(model, t)
        plan = model.plan(B, dev)
        assert plan.fast == "h2" and bool(getattr(plan, "fold_gmax", False)) == fold
        names = [L.name for L in plan.launches]
        assert ("global_max" in names) == fold and ("global_max_concat" in names) != fold
        assert (names.index("spatial_net.xyz_emb") < names.index("upsample")) == fold
        outs[fold] = {k: v.float().clone() for k, v in o.items() if torch.is_tensor(v)}
    assert {"rot", "trans"} <= set(outs[True]) and len(outs[True]) >= 3  # poses + the dense maps
    for k in sorted(outs[True]):
        if outs[True][k].is_floating_point():
            a, b = outs[True][k], outs[False][k]
            d = (a - b).abs().max().item()
            print(f"fold vs concat {k}: max abs diff {d:.3e} (|max| {b.abs().max().item():.3f})")
            assert d <= 1e-4 * max(1.0, b.abs().max().item()), k


@pytest.mark.parametrize("att", ["none", "mul"])
def test_round4_plan_switches_leave_the_outputs_alone(att):
    """cfg.TEST.FUSE_HEAD_OUT (the head's 1x1 output convolution inside the last 3x3 layer's 256x256 epilogue - that layer's activation is
    never written) and cfg.TEST.PNP_H2 (ConvPnPNet on the fp16 matrix pipe in the h2 form, input row written by the glue kernel as an h2
    record, plan-time range proof): the same function as the separate 1x1 launch / the fp32-MFMA ConvPnPNet - dense maps equal up to
    fp32 summation order, poses to 1e-5 - at the batch bench.py times (B = 64) on the well-conditioned weights; and every copy of a crop
    gives the same bits whatever its batch slot (the fused epilogue adds its four partial sums in a fixed order)."""
    import numpy as np

    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    dev = torch.device("cuda:0")
    B = 64
    inp = synth.make_inputs(8, seed=5)
    order = np.random.default_rng(3).permutation(np.repeat(np.arange(8), 8))
    t = {k: torch.from_numpy(np.ascontiguousarray(v[order])).to(dev) for k, v in inp.items()}
    model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention=att, device="cuda"))
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    outs = {}
    for fuse, pnph2 in ((True, True), (False, True), (True, False), (False, False)):
        model.cfg.TEST.FUSE_HEAD_OUT, model.cfg.TEST.PNP_H2 = fuse, pnph2
        with torch.no_grad():
            o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
                      roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
        torch.cuda.synchronize()
        plan = model.plan(B, dev)
        assert plan.fast == "h2" and plan.fused_out == fuse and plan.pnp_h2 == pnph2 and not model.h2_range_exceeded(dev)
        names = [L.name for L in plan.launches]
        assert ("rot_head.out" in names) != fuse
        outs[(fuse, pnph2)] = {k: o[k].clone() for k in ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans")}
        if fuse and pnph2:  # copies of a crop: identical bits in every slot
            for c in range(8):
                slots = np.nonzero(order == c)[0]
                for s in slots[1:]:
                    for k in ("region", "coor_x", "rot", "trans"):
                        assert torch.equal(o[k][slots[0]], o[k][s]), (c, int(s), k)
    ref = outs[(False, False)]
    for key, o in outs.items():
        dm = max((o[k] - ref[k]).abs().max().item() for k in ("mask", "coor_x", "coor_y", "coor_z", "region"))
        dr = max(((o["rot"][b] - ref["rot"][b]).norm() / ref["rot"][b].norm()).item() for b in range(B))
        dt_ = max(((o["trans"][b] - ref["trans"][b]).norm() / ref["trans"][b].norm()).item() for b in range(B))
        print(f"[{att}] FUSE_HEAD_OUT={key[0]} PNP_H2={key[1]} vs both off: maps {dm:.2e}, pose R {dr:.2e} t {dt_:.2e}")
        assert dm <= 2e-5 and dr <= 2e-5 and dt_ <= 2e-5, key
    assert torch.equal(outs[(True, True)]["region"], outs[(True, False)]["region"])  # (the maps do not depend on ConvPnPNet's form)


def test_fused_global_max_leaves_the_outputs_alone():
    """cfg.TEST.FUSE_GLOBAL_MAX (round 5): the point-wise branch's last convolution takes the column-max form - the per-crop channel max
    comes out of its epilogue (64-bit atomic max over the crop's workgroups: order-independent), its 134-MB output is never written.
    Same maxima of the same h2 records: dense maps and poses equal to the separate global-max kernel's BIT FOR BIT at B = 64 (a crop's
    two largest values of a channel never tie with different records on this batch), copies of a crop identical in every slot, and the
    key table is back to zero after every forward."""
    import numpy as np

    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    dev = torch.device("cuda:0")
    B = 64
    inp = synth.make_inputs(8, seed=11)
    order = np.random.default_rng(4).permutation(np.repeat(np.arange(8), 8))
    t = {k: torch.from_numpy(np.ascontiguousarray(v[order])).to(dev) for k, v in inp.items()}
    model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention="mul", device="cuda"))
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    outs = {}
    for fuse in (True, False):
        model.cfg.TEST.FUSE_GLOBAL_MAX = fuse
        model.invalidate_plans()
        for rep in range(2):  # twice: the second forward starts from the keys the first one's decode left
            with torch.no_grad():
                o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
                          roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
            torch.cuda.synchronize()
        plan = model.plan(B, dev)
        names = [L.name for L in plan.launches] + [L.name for L in getattr(plan, "_main_launches", [])]
        assert plan.fast == "h2" and getattr(plan, "fused_gmax", False) == fuse and ("spatial_net.conv3+max" in names) == fuse
        assert ("feat_planes" in plan.bufs) != fuse and not model.h2_range_exceeded(dev)
        if fuse:
            assert int(plan.bufs["gmax_keys"].abs().sum()) == 0
            gm_fused = plan.bufs["gmax_planes"].clone()
        else:
            gm_plain = plan.bufs["gmax_planes"].clone()
        outs[fuse] = {k: o[k].clone() for k in ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans")}
    assert torch.equal(gm_fused.view(torch.int16), gm_plain.view(torch.int16))
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), k
    for c in range(8):
        slots = np.nonzero(order == c)[0]
        for s_ in slots[1:]:
            assert torch.equal(outs[True]["rot"][slots[0]], outs[True]["rot"][s_])


def test_conv_h2_column_max_form_equals_the_max_of_the_written_records():
    """rdpn6d_conv2d_h2_colmax + rdpn6d_h2_colmax_decode (round 5): per group of rows and channel, the h2 record of max(scale * conv + shift)
    - against the ordinary launch's written h2 tensor reduced with torch (the record of the largest reconstructed value; the record is a
    monotonic function of the value, so the two agree value for value), for a 1x1 and a 3x3 layer; the key table is left at zero; a value
    beyond the h2 range (or a NaN) raises the range flag like every other h2 writer."""
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    for (B, H, Cin, N, k) in ((64, 32, 256, 512, 1), (48, 32, 64, 256, 3)):
        g = torch.Generator().manual_seed(B + k)
        x = torch.randn(B, H, H, Cin, generator=g).to(dev)
        w = (torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
        _, ((h2, _), flag) = ops.conv2d_nhwc_h2(x, w, sc, sh, pad=k // 2, want_h2=True)
        rec, keys, fl = ops.conv2d_h2_colmax(x, w, sc, sh, H * H)
        torch.cuda.synchronize()
        h = h2.view(B, H * H, N // 32, 2, 32).float()
        want = (h[:, :, :, 0] + h[:, :, :, 1]).amax(dim=1)
        got = rec[:, :, 0].float() + rec[:, :, 1].float()
        assert torch.equal(got, want), (B, k, (got - want).abs().max().item())
        assert int(keys.abs().sum()) == 0 and int(fl) == 0 and int(flag) == 0
    xb = x.clone()
    xb[3, 5, 5, :] = 3.0e4  # one pixel far beyond +-4094 after the convolution
    _, _, fl = ops.conv2d_h2_colmax(xb, w, sc, sh, H * H)
    torch.cuda.synchronize()
    assert int(fl) == 1
    xb[3, 5, 5, 0] = float("nan")
    _, _, fl = ops.conv2d_h2_colmax(xb, w, sc, sh, H * H)
    torch.cuda.synchronize()
    assert int(fl) == 1
