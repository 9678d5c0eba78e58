"""Phases of the fused stem kernel (probe build only): RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force && python tools/probe_stem.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr, pack_stem_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
lib.rdpn6d_debug_stem_probe.argtypes = [ctypes.c_void_p]
B, R = 64, 256
probe = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
_lib.check(lib.rdpn6d_debug_stem_probe(_ptr(probe)))
x = torch.rand(B, 6, R, R, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) / 12
wh, inv = pack_stem_h2_weight(w)
sc = (torch.rand(64, device=dev) + 0.5) * inv; sh = torch.randn(64, device=dev) * 0.3
y = torch.empty(B * 64 * 64, 2, 2, 32, dtype=torch.float16, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3): _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(x), B, 6, R, _ptr(wh), _ptr(sc), _ptr(sh), _ptr(y), _ptr(flag), st))
torch.cuda.synchronize(); probe.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(x), B, 6, R, _ptr(wh), _ptr(sc), _ptr(sh), _ptr(y), _ptr(flag), st)); e1.record()
torch.cuda.synchronize()
p = probe.cpu().numpy().reshape(-1, 4); p = p[p[:, 3] > 0]
print(f"stem+pool B={B}: {e0.elapsed_time(e1)*1e3:.1f} us, {len(p)//4} workgroups; per wave (cycles): patch load + split {p[:,0].mean():.0f} | MFMA phase {p[:,1].mean():.0f} | epilogue + pool {p[:,2].mean():.0f}")
