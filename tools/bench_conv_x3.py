"""Hot-loop timing of the head 3x3 layer (256 -> 256 @ 64x64, B=64): fp32-MFMA kernel vs the bf16x3 kernel."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, C, k = int(os.environ.get("B", 64)), 64, 256, 3
x = torch.randn(B, H, H, C, device=dev)
w = torch.randn(C, C, k, k, device=dev) / (C * 9) ** 0.5
wp32 = pack_conv_weight(w)
xp, wp = ops.split_bf16x3(x), ops.split_bf16x3(wp32)
y = torch.empty(B, H, H, C, device=dev)
yp = torch.empty(3, y.numel(), dtype=torch.bfloat16, device=dev)
def desc(xt, wt):
    d = _lib.ConvDesc()
    d.x, d.w, d.y = _ptr(xt), _ptr(wt), _ptr(y)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, C, C, H, H, 1
    taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    d.ntaps = 9
    for t, (dy, dx) in enumerate(taps): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act = C, C, H, H, 1, 1, C, 1
    return d
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
d32, dx3 = desc(x, wp32), desc(xp, wp)
gf = 2.0 * B * H * H * C * C * 9 / 1e9
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t32 = timeit(lambda: _lib.check(lib.rdpn6d_conv2d_f32(ctypes.byref(d32), st)))
tx3 = timeit(lambda: _lib.check(lib.rdpn6d_conv2d_bf16x3(ctypes.byref(dx3), xp.shape[1], wp.shape[1], None, 0, st)))
tx3p = timeit(lambda: _lib.check(lib.rdpn6d_conv2d_bf16x3(ctypes.byref(dx3), xp.shape[1], wp.shape[1], _ptr(yp), yp.shape[1], st)))
print(f"B={B}: fp32-MFMA {t32:8.1f} us ({gf/t32*1e3:6.1f} TF/s) | bf16x3 {tx3:8.1f} us ({gf/tx3*1e3:6.1f} TF/s fp32-equivalent, "
      f"{6*gf/tx3*1e3:6.1f} TF/s of bf16 MFMA) | bf16x3 + planes out {tx3p:8.1f} us")
