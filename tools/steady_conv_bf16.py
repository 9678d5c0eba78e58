"""Steady-state loop of one bf16 conv shape (for rocprofv3 --pmc passes): head 3x3 256->256 @64x64, B=64."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _pad_to, _ptr, pack_conv_weight
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, Cout = 64, 64, 256
Cin, k = int(os.environ.get("CIN", 256)), int(os.environ.get("KSIZE", 3))
tiles = tuple(int(v) for v in os.environ.get("TILE", "0,0").split(","))
x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
wp = pack_conv_weight(torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
y = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
d = _lib.ConvDesc()
d.x, d.w, d.y = _ptr(x), _ptr(wp), _ptr(y)
d.B, d.H, d.W, d.Cin, d.in_cs = B, H, H, Cin, Cin
d.Ho, d.Wo, d.stride = H, H, 1
taps = [(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]
d.ntaps = k * k
for t, (dy, dx) in enumerate(taps): d.dy[t], d.dx[t] = dy, dx
d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act = Cout, Cout, H, H, 1, 1, Cout, 1
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.rdpn6d_conv_bf16_force_tile(*tiles)
if "CHUNK" in os.environ:
    lib.rdpn6d_conv_bf16_force_chunk.restype = None
    lib.rdpn6d_conv_bf16_force_chunk(int(os.environ["CHUNK"]))
if "TIME" in os.environ:
    for _ in range(5): lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"Cin {Cin} k {k} tile {tiles} chunk {os.environ.get('CHUNK', 'auto')}: {us:8.1f} us  {2.0*B*H*H*Cout*k*k*Cin/us/1e6:7.1f} TF/s")
for _ in range(int(os.environ.get("REPS", 30))): _lib.check(lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st))
torch.cuda.synchronize()
