"""Prologue / K loop / epilogue of the 256x256 eight-phase h2 kernel on a head layer (probe build only):
    RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force && python tools/probe_h2_8ph.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight, pack_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
lib.rdpn6d_debug_h2_probe.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("B", 64))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
probe = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
_lib.check(lib.rdpn6d_debug_h2_probe(_ptr(probe)))
H, C = 64, 256
x = torch.randn(B, H, H, C, device=dev).clamp(min=0)
w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
wp32 = pack_conv_weight(w); wh, inv = pack_h2_weight(wp32)
xh, _ = ops.split_h2(x); yh = torch.empty_like(xh)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
d = _lib.ConvDesc()
d.x, d.w, d.scale = _ptr(xh), _ptr(wh), _ptr(inv)
d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, C, C, H, H, 1
d.ntaps = 9
for t, (dy, dx) in enumerate([(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]): d.dy[t], d.dx[t] = dy, dx
d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act, d.res_cs = C, wp32.shape[0], H, H, 1, 1, C, 1, C
for _ in range(3):
    _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), None, _ptr(flag), st))
torch.cuda.synchronize(); probe.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), None, _ptr(flag), st)); e1.record()
torch.cuda.synchronize()
p = probe.cpu().numpy().reshape(-1, 8)
p = p[p[:, 6] > 0].reshape(-1, 8, 8)
wg = p[:, 0]  # wave 0 of every workgroup
rs, re = wg[:, 4], wg[:, 5]
t0 = rs.min()
order = np.argsort(rs)
print(f"head layer B={B}: {e0.elapsed_time(e1)*1e3:.1f} us, {len(wg)} workgroups, {int(wg[0, 6])} K-tiles")
print(f"  per workgroup (shader cycles, mean): set-up + prologue {p[:, :, 0].mean():7.0f} | K loop {p[:, :, 1].mean():8.0f} | epilogue to last store issued {p[:, :, 2].mean():7.0f} | stores done {p[:, :, 3].mean():6.0f}")
dur = (re - rs) / 100.0
print(f"  100 MHz clock: workgroup duration us p10/p50/p90 {np.percentile(dur, [10, 50, 90]).round(1).tolist()}; starts (us after the first), sorted, every 128th: {((rs[order][::128] - t0) / 100.0).round(1).tolist()}")
clk = (p[:, 0, 0] + p[:, 0, 1] + p[:, 0, 2] + p[:, 0, 3]) / (dur * 1e-6) / 1e9
print(f"  shader clock over a workgroup's life: {np.median(clk):.2f} GHz")
