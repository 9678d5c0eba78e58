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
    y32 = ops.conv2d_nhwc(x, w, sc, sh, stride=stride, pad=k // 2, act=1, residual=res, out_f32=True)
    y16 = ops.conv2d_nhwc(x, w, sc, sh, stride=stride, pad=k // 2, act=1, residual=res) if Cout % 8 == 0 else None
    e32 = ((y32 - ref).abs().max() / ref.abs().max()).item()
    e16 = ((y16.float() - ref.bfloat16().float()).abs().max() / ref.abs().max()).item() if y16 is not None else -1
    print(f"check {name:34s} f32-out rel err {e32:.2e}   bf16-out vs rounded ref {e16:.2e}")
    assert e32 < 2e-5 and e16 < 1e-2
def bench(name, H, Cin, Cout, k, stride, tiles=None, reps=20):
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).bfloat16()
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    from rdpn6d_amd.gdrn import _pad_to
    out = torch.empty(B, Ho, Ho, _pad_to(Cout, 8), device=dev, dtype=torch.bfloat16)
    if tiles: lib.rdpn6d_conv_bf16_force_tile(*tiles)
    f = lambda: ops.conv2d_nhwc(x, w, None, None, stride=stride, pad=k // 2, act=1, out=out)
    for _ in range(3): f()
    # time only the kernel: re-launch the prepared descriptor through ops is python-heavy, so use events around reps
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    lib.rdpn6d_conv_bf16_force_tile(0, 0)
    print(f"{name:34s} tiles {str(tiles):10s} {best*1e3:9.1f} us  {fl/best/1e9:7.1f} TF/s  ({fl/best/1e9/2500*100:5.1f}% of bf16 MFMA peak)")
for s in SHAPES: check(*s)
for s in SHAPES: bench(*s)
if os.environ.get("SWEEP"):
    for s in SHAPES[:6]:
        for t in ((128, 128), (128, 64), (64, 128), (64, 64)):
            if max(64, s[3]) % t[1] == 0: bench(*s, tiles=t)
