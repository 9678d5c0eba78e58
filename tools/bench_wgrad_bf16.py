"""GPU-box check + microbenchmark of the 16-bit weight-gradient kernel on the training shapes (B=32): compares with a torch fp32
einsum over the same bf16-rounded operands, then times.  RDPN6D_WGRAD_WIDE=0: without the 256 x 128 tile."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr
lib = _lib.load(); dev = torch.device("cuda:0"); B = int(os.environ.get("B", 32))
SHAPES = [("head 3x3 256->256 @64", 64, 256, 256, 3), ("layer2 3x3 128 @32", 32, 128, 128, 3), ("layer3 3x3 256 @16", 16, 256, 256, 3),
          ("layer4 3x3 512 @8", 8, 512, 512, 3), ("1x1 512->256 @32", 32, 512, 256, 1), ("head-like 3x3 256->512 @32", 32, 256, 512, 3)]
def run(name, H, Cin, Cout, k, Bn, check):
    x = torch.randn(Bn, H, H, Cin, device=dev).bfloat16(); dy = (torch.randn(Bn, H, H, Cout, device=dev) / 64).bfloat16()
    taps = [(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]
    tdy = (ctypes.c_int * 9)(*[t[0] for t in taps] + [0] * (9 - len(taps))); tdx = (ctypes.c_int * 9)(*[t[1] for t in taps] + [0] * (9 - len(taps)))
    out = torch.empty(Cout, k * k, Cin, device=dev)
    part = torch.empty(int(lib.rdpn6d_wgrad_scratch_floats(Bn, H, H, Cout, Cin, k * k)), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: _lib.check(lib.rdpn6d_wgrad_bf16(_ptr(dy), Cout, 0, Cout, Cout, _ptr(x), Cin, 0, Cin, Cin, Bn, H, H, H, H, 1, k * k, tdy, tdx,
                                                 _ptr(out), _ptr(part), st))
    f()
    if check:
        xp = F.pad(x.float(), (0, 0, k // 2, k // 2, k // 2, k // 2))
        ref = torch.stack([torch.einsum("bhwo,bhwi->oi", dy.float(), xp[:, k // 2 + a:k // 2 + a + H, k // 2 + b:k // 2 + b + H]) for a, b in taps], dim=1)
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        print(f"check {name:28s} B={Bn} rel err {err:.2e}")
        assert err < 2e-5, err
        return
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10; fl = 2.0 * Bn * H * H * Cout * k * k * Cin
    print(f"{name:28s} {ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF/s ({fl/ms/1e9/2500*100:5.1f}% of the 16-bit MFMA peak)")
for s in SHAPES: run(*s, 3, True)
for s in SHAPES: run(*s, B, False)
