"""Hot-loop timing of the head layer's weight gradient (256x256x3x3 @64x64, B=32): fp32-MFMA vs bf16 vs bf16x3 kernels."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr
lib = _lib.load(); dev = torch.device("cuda:0")
B, H, C, k = int(os.environ.get("B", 32)), 64, 256, 3
dy, x = torch.randn(B, H, H, C, device=dev), torch.randn(B, H, H, C, device=dev)
dyp, xp = ops.split_bf16x3(dy), ops.split_bf16x3(x)
dyb, xb = dy.bfloat16(), x.bfloat16()
taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
tdy = (ctypes.c_int * 9)(*[t[0] for t in taps]); tdx = (ctypes.c_int * 9)(*[t[1] for t in taps])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scr = torch.empty(int(lib.rdpn6d_wgrad_scratch_floats(B, H, H, C, C, 9)), device=dev)
out = torch.empty(C, 9, C, device=dev)
gf = 2.0 * B * H * H * C * C * 9 / 1e9
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t32 = timeit(lambda: _lib.check(lib.rdpn6d_wgrad_f32(_ptr(dy), C, 0, C, _ptr(x), C, 0, C, B, H, H, H, H, 1, 9, tdy, tdx, _ptr(out), _ptr(scr), st)))
t16 = timeit(lambda: _lib.check(lib.rdpn6d_wgrad_bf16(_ptr(dyb), C, 0, C, C, _ptr(xb), C, 0, C, C, B, H, H, H, H, 1, 9, tdy, tdx, _ptr(out), _ptr(scr), st)))
tx3 = timeit(lambda: _lib.check(lib.rdpn6d_wgrad_bf16x3_strided(_ptr(dyp), dyp.shape[1], C, 0, C, C, _ptr(xp), xp.shape[1], C, 0, C, C, B, H, H, H, H,
                                                                 1, 9, tdy, tdx, _ptr(out), 9 * C, C, 1, C, C, _ptr(scr), st)))
print(f"B={B}: fp32-MFMA {t32:7.1f} us ({gf/t32*1e3:6.1f} TF/s) | bf16 {t16:7.1f} us ({gf/t16*1e3:6.1f}) | bf16x3 {tx3:7.1f} us ({gf/tx3*1e3:6.1f} TF/s fp32-equivalent)")
