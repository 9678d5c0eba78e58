"""Hot-loop timing of trunk- and head-shaped 3x3 layers at B=64 on the two-plane fp16 (h2) kernels: h2 tensor -> h2 tensor
(+ h2 residual), as the layers run inside the plan.  Env: RDPN6D_H2_NST=2|3, RDPN6D_H2_TILE=bm,bn (read once per process)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight, pack_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"NST={os.environ.get('RDPN6D_H2_NST', 'auto')} TILE={os.environ.get('RDPN6D_H2_TILE', 'auto')}")
for name, H, C in (("layer1", 64, 64), ("layer2", 32, 128), ("layer3", 16, 256), ("layer4", 8, 512), ("head", 64, 256)):
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
    wp32 = pack_conv_weight(w)
    wh, inv = pack_h2_weight(wp32)
    xh, _ = ops.split_h2(x)
    rh, _ = ops.split_h2(torch.randn(B, H, H, C, device=dev))
    yh = torch.empty_like(xh)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale = _ptr(xh), _ptr(wh), _ptr(inv)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, C, C, H, H, 1
    d.ntaps = 9
    for t, (dy, dx) in enumerate([(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act, d.res_cs = C, wp32.shape[0], H, H, 1, 1, C, 1, C
    gf = 2.0 * B * H * H * C * C * 9 / 1e9
    mb = 3 * B * H * H * C * 4 / 1e6
    t = timeit(lambda: _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), st)))
    print(f"{name} ({H}x{H}x{C}, kernel {lib.rdpn6d_conv_h2_kernel_for(ctypes.byref(d))}): {t:7.1f} us  {gf/t*1e3:6.1f} TF/s fp32-equivalent "
          f"({3*gf/t*1e3:6.0f} issued)  {mb/t*1e3/1e3:5.2f} TB/s of activations (in + res + out)")
