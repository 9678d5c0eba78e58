import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _pad_to, _pad_vec, _ptr, pack_conv_weight
lib, dev = _lib.load(), torch.device("cuda:0")
# 1. cast kernel
x = torch.randn(4, 64, device=dev)
y = torch.zeros(4, 64, dtype=torch.float16, device=dev)
_lib.check(lib.rdpn6d_cast_f32_fp16(_ptr(x), 64, 0, 64, _ptr(y), 64, 4, None)); torch.cuda.synchronize()
print("cast f32->fp16 max err", (y.float() - x.half().float()).abs().max().item())
# 2. maxpool fp16
xm = torch.randn(1, 8, 8, 64, device=dev).half(); ym = torch.zeros(1, 4, 4, 64, dtype=torch.float16, device=dev)
_lib.check(lib.rdpn6d_maxpool3x3s2_fp16(_ptr(xm), 1, 8, 8, 64, _ptr(ym), None)); torch.cuda.synchronize()
ref = torch.nn.functional.max_pool2d(xm.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
print("maxpool fp16 max err", (ym.float() - ref).abs().max().item())
for (B, H, Cin, Cout, k, stride, use_res, act) in [(2, 16, 64, 128, 3, 1, True, 1), (2, 16, 64, 128, 3, 1, False, 1), (2, 16, 64, 128, 3, 1, True, 0), (2, 16, 64, 128, 1, 1, True, 0), (1, 64, 256, 256, 3, 1, True, 1)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, H, Cin, generator=g).half().to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half().float()
    wp = pack_conv_weight(w.to(dev), cin_pad=_pad_to(Cin, 32))
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, generator=g).half().to(dev) if use_res else None
    def desc(xt, wt, yt, rt):
        d = _lib.ConvDesc()
        d.x, d.w, d.res, d.y = _ptr(xt), _ptr(wt), _ptr(rt), _ptr(yt)
        d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, Cin, Cin, Ho, Ho, stride
        taps = [(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]
        d.ntaps = len(taps)
        for t, (dy, dx) in enumerate(taps): d.dy[t], d.dx[t] = dy, dx
        d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.res_cs, d.act = Cout, wp.shape[0], Ho, Ho, 1, 1, Cout, Cout, act
        return d
    y16 = torch.empty(B, Ho, Ho, Cout, dtype=torch.float16, device=dev); y32 = torch.empty(B, Ho, Ho, Cout, device=dev)
    yf = torch.empty(B, Ho, Ho, Cout, device=dev)
    d16 = desc(x, wp.half(), y16, res); _lib.check(lib.rdpn6d_conv2d_fp16(ctypes.byref(d16), 0, None))
    d16f = desc(x, wp.half(), yf, res.float() if use_res else None); _lib.check(lib.rdpn6d_conv2d_fp16(ctypes.byref(d16f), 1, None))
    d32 = desc(x.float(), wp, y32, res.float() if use_res else None); _lib.check(lib.rdpn6d_conv2d_f32(ctypes.byref(d32), None))
    torch.cuda.synchronize()
    bm, bn = ctypes.c_int(), ctypes.c_int(); lib.rdpn6d_conv_fp16_tile_for(ctypes.byref(d16), ctypes.byref(bm), ctypes.byref(bn))
    e = (y16.float() - y32).abs(); ef = (yf - y32).abs()
    print((B, H, Cin, Cout, k, stride, use_res, act), "tile", bm.value, bn.value, "fp16-out err", e.max().item(), "f32-out err", ef.max().item(),
          "bad frac", (e > 0.05).float().mean().item(), "first bad idx", (e > 0.05).nonzero()[:3].tolist())
