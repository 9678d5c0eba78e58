"""GPU-box microbenchmark of the conv kernel family on the shapes of the forward path (B=64)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight, _pad_to

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))
SHAPES = [  # name, H, Cin, Cout, k, stride
    ("head 3x3 256->256 @64", 64, 256, 256, 3, 1),
    ("layer1 3x3 64->64 @64", 64, 64, 64, 3, 1),
    ("layer2 3x3 128->128 @32", 32, 128, 128, 3, 1),
    ("layer3 3x3 256->256 @16", 16, 256, 256, 3, 1),
    ("layer4 3x3 512->512 @8", 8, 512, 512, 3, 1),
    ("convT phase(4 taps) 1024->256 @32", 32, 1024, 256, 2, 1),
    ("pointnet 1x1 512->64 @32", 32, 512, 64, 1, 1),
    ("head out 1x1 256->37 @64", 64, 256, 37, 1, 1),
]
def run(name, H, Cin, Cout, k, stride, tiles=None, reps=10):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    wp = pack_conv_weight(w)
    pad = k // 2 if k != 2 else 0
    Ho = (H + 2 * pad - k) // stride + 1 if k != 2 else H
    y = torch.empty(B, Ho, Ho, Cout if Cout % 4 == 0 else _pad_to(Cout, 4), device=dev)
    sc = torch.ones(wp.shape[0], device=dev); sh = torch.zeros(wp.shape[0], device=dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.y = _ptr(x), _ptr(wp), _ptr(sc), _ptr(sh), _ptr(y)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, H, Cin, Cin, 0
    d.Ho, d.Wo, d.stride = Ho, Ho, stride
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = Cout, wp.shape[0], Ho, Ho
    d.osy = d.osx = 1; d.out_cs = y.shape[-1]; d.act = 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if tiles: lib.rdpn6d_conv_force_tile(*tiles)
    bm, bn = ctypes.c_int(), ctypes.c_int()
    lib.rdpn6d_conv_tile_for(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn))
    for _ in range(2): _lib.check(lib.rdpn6d_conv2d_f32(ctypes.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.rdpn6d_conv2d_f32(ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Ho * Ho * Cout * len(taps) * Cin
    lib.rdpn6d_conv_force_tile(0, 0)
    print(f"{name:38s} tile {bm.value:3d}x{bn.value:3d}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF/s  ({fl/ms/1e9/157.3*100:5.1f}% of fp32 MFMA peak)")
    return ms
for ti in ([int(os.environ["TAP_INNER"])] if "TAP_INNER" in os.environ else [1, 0]):
    lib.rdpn6d_conv_set_tap_inner.restype = None
    lib.rdpn6d_conv_set_tap_inner(ti)
    print("tap_inner =", ti)
    for s in SHAPES:
        run(*s)
if os.environ.get("SWEEP"):
    for s in SHAPES[:5]:
        for t in ((128, 128), (128, 64), (64, 128), (64, 64)):
            if _pad_to(s[3], 64) % t[1] == 0: run(*s, tiles=t)
