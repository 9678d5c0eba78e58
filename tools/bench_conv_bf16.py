"""GPU-box check + microbenchmark of the bf16 conv kernel: compares with the fp32 HIP kernel fed the same
bf16-rounded operands (differences = fp32 summation order only), then times the forward-path shapes."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))
SHAPES = [  # name, H, Cin, Cout, k, stride
    ("head 3x3 256->256 @64", 64, 256, 256, 3, 1),
    ("layer1 3x3 64->64 @64", 64, 64, 64, 3, 1),
    ("layer2 3x3 128->128 @32", 32, 128, 128, 3, 1),
    ("layer2 ds 3x3s2 64->128 @64", 64, 64, 128, 3, 2),
    ("layer3 3x3 256->256 @16", 16, 256, 256, 3, 1),
    ("layer4 3x3 512->512 @8", 8, 512, 512, 3, 1),
    ("1x1 512->64 @32", 32, 512, 64, 1, 1),
    ("1x1 96->64 @64 (RB=64)", 64, 96, 64, 1, 1),
    ("head out 1x1 256->37 @64", 64, 256, 37, 1, 1),
    ("convT phase 2x2 1024->256 @32", 32, 1024, 256, 2, 1),
]
torch.manual_seed(0)
def check(name, H, Cin, Cout, k, stride):
    Bc = 3
    x = torch.randn(Bc, H, H, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    sc = torch.rand(Cout, device=dev) + 0.5
    sh = torch.randn(Cout, device=dev)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(Bc, Ho, Ho, Cout, device=dev).bfloat16() if Cout % 8 == 0 else None
    ref = ops.conv2d_nhwc(x.float(), w.float(), sc, sh, stride=stride, pad=k // 2, act=1,
                          residual=res.float() if res is not None else None)
    y32 = ops.conv2d_nhwc(x, w, sc, sh, stride=stride, pad=k // 2, act=1, residual=res.float() if res is not None else None, out_f32=True)
    y16 = ops.conv2d_nhwc(x, w, sc, sh, stride=stride, pad=k // 2, act=1, residual=res) if Cout % 8 == 0 else None
    e32 = ((y32 - ref).abs().max() / ref.abs().max()).item()
    e16 = ((y16.float() - ref.bfloat16().float()).abs().max() / ref.abs().max()).item() if y16 is not None else -1
    print(f"check {name:34s} f32-out rel err {e32:.2e}   bf16-out vs rounded ref {e16:.2e}")
    assert e32 < 2e-5 and e16 < 1e-2
def bench(name, H, Cin, Cout, k, stride, tiles=None, reps=20):
    from rdpn6d_amd.gdrn import _pad_to, _ptr, pack_conv_weight
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    wp = pack_conv_weight(w, cin_pad=_pad_to(Cin, 32)).bfloat16()
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    y = torch.empty(B, Ho, Ho, _pad_to(Cout, 8), device=dev, dtype=torch.bfloat16)
    d = _lib.ConvDesc()
    d.x, d.w, d.y = _ptr(x), _ptr(wp), _ptr(y)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.in_co = B, H, H, Cin, Cin, 0
    d.Ho, d.Wo, d.stride = Ho, Ho, stride
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    d.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW = Cout, wp.shape[0], Ho, Ho
    d.osy = d.osx = 1; d.out_cs = y.shape[-1]; d.act = 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if tiles: lib.rdpn6d_conv_bf16_force_tile(*tiles)
    bm, bn = ctypes.c_int(), ctypes.c_int()
    lib.rdpn6d_conv_bf16_tile_for(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn))
    for _ in range(3): _lib.check(lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    extra = ""
    if os.environ.get("STATS") and Cout % 64 == 0:  # the training forward's instantiation: the epilogue also writes BatchNorm partial sums
        d.scale = d.shift = None
        d.act = 0
        stats = torch.empty(((B * Ho * Ho + 63) // 64) * 2 * Cout * 2 + 1024, device=dev, dtype=torch.float64)
        rows = ctypes.c_int()
        for _ in range(3): _lib.check(lib.rdpn6d_conv2d_bf16_bnstats(ctypes.byref(d), _ptr(stats), 0, ctypes.byref(rows), st))
        e0.record()
        for _ in range(reps): lib.rdpn6d_conv2d_bf16_bnstats(ctypes.byref(d), _ptr(stats), 0, ctypes.byref(rows), st)
        e1.record(); torch.cuda.synchronize()
        extra = f"   with BatchNorm partial sums: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us ({rows.value} rows)"
    if os.environ.get("FLUSH"):  # every launch behind a fill of FLUSH MB (cold L2s; 512: cold Infinity Cache too), timed one by one
        junk = torch.empty(int(os.environ["FLUSH"]) << 20, device=dev, dtype=torch.uint8)
        tot = 0.0
        for i in range(10):
            junk.fill_(i)
            e0.record()
            lib.rdpn6d_conv2d_bf16(ctypes.byref(d), 0, st)
            e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        extra += f"   behind a {os.environ['FLUSH']}-MB fill: {tot / 10 * 1e3:7.1f} us"
        del junk
    lib.rdpn6d_conv_bf16_force_tile(0, 0)
    print(f"{name:34s} tile {bm.value:3d}x{bn.value:3d} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF/s  ({fl/ms/1e9/2500*100:5.1f}% of bf16 MFMA peak){extra}")
if os.environ.get("NST"): lib.rdpn6d_conv_bf16_force_stages(int(os.environ["NST"]))  # profiling: LDS stages of the 4-wave tiles
for s in SHAPES: check(*s)
for s in SHAPES: bench(*s)
if os.environ.get("SWEEP"):
    for s in SHAPES[:6]:
        for t in ((256, 128), (128, 128), (128, 64), (64, 128), (64, 64)):
            if max(64, s[3]) % t[1] == 0: bench(*s, tiles=t)
