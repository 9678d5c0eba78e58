"""Parts of a step of the 8-wave ping-pong kernel (conv_igemm_h2_pp.hip), probe build only:
    RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force && python tools/probe_h2_pp.py
per wave, shader cycles summed over the K loop: L part issue (fragment reads, addresses, PL DMA pieces) | its waits | barrier |
M part (MFMAs + PM pieces) | its wait | barrier."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight, pack_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
lib.rdpn6d_debug_h2pp_probe.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("B", 64))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
probe = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
_lib.check(lib.rdpn6d_debug_h2pp_probe(_ptr(probe)))
for name, H, C in (("layer2", 32, 128), ("layer3", 16, 256)):
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
    wp32 = pack_conv_weight(w); wh, inv = pack_h2_weight(wp32)
    xh, _ = ops.split_h2(x); rh, _ = ops.split_h2(torch.randn(B, H, H, C, device=dev)); yh = torch.empty_like(xh)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale = _ptr(xh), _ptr(wh), _ptr(inv)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, C, C, H, H, 1
    d.ntaps = 9
    for t, (dy, dx) in enumerate([(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act, d.res_cs = C, wp32.shape[0], H, H, 1, 1, C, 1, C
    for _ in range(3):
        _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), st))
    torch.cuda.synchronize(); probe.zero_()
    _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), st))
    torch.cuda.synchronize()
    p = probe.cpu().numpy().reshape(-1, 16)
    p = p[p[:, 7] > 0]
    nk = int(p[0, 7])
    for g in (0, 1):
        q = p.reshape(-1, 8, 16)[:, 4 * g:4 * g + 4].reshape(-1, 16)
        m = q[:, :6].mean(0) / nk
        print(f"{name} group {g}: per chunk  L issue {m[0]:5.0f}  L waits {m[1]:5.0f}  barrier {m[2]:5.0f} | M {m[3]:5.0f}  M wait {m[4]:5.0f}  barrier {m[5]:5.0f}  = {m.sum():6.0f} cycles; loop {q[:, 6].mean():8.0f} cycles, {nk} chunks, {len(q)} waves\n"
              f"      outside the loop: set-up + prologue {q[:, 8].mean():6.0f} | drain + re-align {q[:, 9].mean():6.0f} | residual loads + barrier {q[:, 10].mean():6.0f} | "
              f"scale/shift + transposes + stores issued {q[:, 11].mean():6.0f} | stores done {q[:, 12].mean():6.0f}")
