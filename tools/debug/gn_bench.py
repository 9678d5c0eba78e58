"""GroupNorm(32,128)+ReLU kernels of ConvPnPNet alone: in-place fp32 form vs h2-output form, hot and behind a 512-MB fill."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr

lib = _lib.load()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for B, HW in ((64, 1024), (64, 256), (64, 64), (32, 1024)):
    C, G = 128, 32
    x = torch.randn(B, HW, C, device=dev)
    ga, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    y = torch.empty(B * HW * C * 2, dtype=torch.float16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fns = {"fp32 in place": lambda: lib.rdpn6d_groupnorm_relu_f32(_ptr(x), B, HW, C, G, _ptr(ga), _ptr(be), st),
           "h2 output": lambda: lib.rdpn6d_groupnorm_relu_h2(_ptr(x), B, HW, C, G, _ptr(ga), _ptr(be), _ptr(y), _ptr(flag), st)}
    for name, fn in fns.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        hot = e0.elapsed_time(e1) / 50 * 1e3
        tot = 0.0
        for i in range(10):
            junk.fill_(i)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print(f"B {B} HW {HW:5d}  {name:14s} hot {hot:7.1f} us   cold {tot / 10 * 1e3:7.1f} us   ({B * HW * C * 8 / 1e6:.1f} MB moved)")
