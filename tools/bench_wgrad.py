"""GPU-box microbenchmark of the wgrad kernel on the training shapes (B=32)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr
lib = _lib.load(); dev = torch.device("cuda:0"); B = int(os.environ.get("B", 32))
SHAPES = [("head 3x3 256->256 @64", 64, 256, 256, 3, 1), ("layer1 3x3 64 @64", 64, 64, 64, 3, 1), ("layer2 3x3 128 @32", 32, 128, 128, 3, 1),
          ("layer3 3x3 256 @16", 16, 256, 256, 3, 1), ("layer4 3x3 512 @8", 8, 512, 512, 3, 1), ("fc1 8192->1024", 1, 8192, 1024, 1, 1)]
for name, H, Cin, Cout, k, s in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    taps = [(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]
    tdy = (ctypes.c_int * 9)(*[t[0] for t in taps] + [0] * (9 - len(taps))); tdx = (ctypes.c_int * 9)(*[t[1] for t in taps] + [0] * (9 - len(taps)))
    out = torch.empty(Cout, k * k, Cin, device=dev)
    part = torch.empty(int(lib.rdpn6d_wgrad_scratch_floats(B, H, H, Cout, Cin, k * k)), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: _lib.check(lib.rdpn6d_wgrad_f32(_ptr(dy), Cout, 0, Cout, _ptr(x), Cin, 0, Cin, B, H, H, H, H, s, k * k, tdy, tdx, _ptr(out), _ptr(part), st))
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10; fl = 2.0 * B * H * H * Cout * k * k * Cin
    print(f"{name:28s} {ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF/s ({fl/ms/1e9/157.3*100:5.1f}%)  splits {part.numel()//out.numel()}")
