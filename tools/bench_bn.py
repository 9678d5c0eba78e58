"""Micro-benchmark of the training BatchNorm kernels (csrc/train_norm.hip) on the shapes of the B=32 training step:
achieved HBM GB/s of the statistics pass, the apply pass and the two backward passes.  python tools/bench_bn.py [bf16|f32]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rdpn6d_amd import _lib  # noqa: E402
from rdpn6d_amd.gdrn import _ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tdt = torch.bfloat16 if dt == "bf16" else torch.float32
es = 2 if dt == "bf16" else 4
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
scratch = torch.empty(512 * 1024 * 2 + 4096, dtype=torch.float64, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, C in [("stem", 32 * 128 * 128, 64), ("layer1", 32 * 64 * 64, 64), ("layer2", 32 * 32 * 32, 128), ("layer3", 32 * 16 * 16, 256),
                   ("layer4", 32 * 8 * 8, 512), ("head", 32 * 64 * 64, 256), ("pn1024", 32 * 32 * 32, 1024)]:
    x = torch.randn(M, C, device=dev).to(tdt)
    dy = torch.randn(M, C, device=dev).to(tdt)
    res = torch.randn(M, C, device=dev).to(tdt)
    y, dx, dres = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    mean, istd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ga, be, dga, dbe = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    f_stats, f_apply, f_bwd = (getattr(lib, f"rdpn6d_bn_{n}_{dt}") for n in ("train_stats", "apply", "backward"))
    t_s = timeit(lambda: f_stats(_ptr(x), M, C, C, 0, 1e-5, 0.1, _ptr(mean), _ptr(istd), _ptr(rm), _ptr(rv), _ptr(scratch), st()))
    t_a = timeit(lambda: f_apply(_ptr(x), C, 0, _ptr(mean), _ptr(istd), _ptr(ga), _ptr(be), _ptr(res), C, 0, _ptr(y), C, 0, M, C, 1, st()))
    t_b = timeit(lambda: f_bwd(_ptr(x), C, 0, _ptr(dy), C, 0, _ptr(y), C, 0, _ptr(mean), _ptr(istd), _ptr(ga), _ptr(dga), _ptr(dbe),
                               _ptr(dx), C, 0, _ptr(dres), C, 0, M, C, 1, _ptr(scratch), st()))
    b = M * C * es / 1e3  # KB... bytes / 1e3 -> us * GB/s
    print(f"{name:8s} M={M:7d} C={C:5d} tensor {M*C*es/1e6:6.1f} MB | stats {t_s:6.1f} us {b/t_s:6.0f} GB/s | apply(+res) {t_a:6.1f} us "
          f"{3*b/t_a:6.0f} GB/s | backward(3 launches, 6R+2W) {t_b:6.1f} us {8*b/t_b:6.0f} GB/s")
