"""probe build (RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force): where a tile of conv3x3_c64_h2_kernel goes, in shader cycles"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rdpn6d_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(_lib.LIB_PATH)
probe = torch.zeros(8 * 4 * 8, dtype=torch.int64, device=dev)
lib.rdpn6d_debug_c64_probe(ctypes.c_void_p(probe.data_ptr()))
B = 64
x = torch.randn(B, 64, 64, 64, device=dev).relu_()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.06
xh, _ = ops.split_h2(x)
for res in (False, True):
    rh = (xh, (B, 64, 64, 64)) if res else None
    for _ in range(3):
        ops.conv2d_nhwc_h2((xh, (B, 64, 64, 64)), w, stride=1, pad=1, act=1, want_h2=True, residual_h2=rh, c64=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_nhwc_h2((xh, (B, 64, 64, 64)), w, stride=1, pad=1, act=1, want_h2=True, residual_h2=rh, c64=True)
    e1.record()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.conv2d_nhwc_h2((xh, (B, 64, 64, 64)), w, stride=1, pad=1, act=1, want_h2=True, residual_h2=rh, split_k=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"  (tile kernel through the same helper: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us)")
    p = probe.cpu().reshape(8, 4, 8).numpy()
    print(f"residual {res}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (incl. the host-side weight pack of the test helper)")
    for tile in (0, 3, 6):
        for wv in (0, 2):
            s = p[tile, wv]
            nxt = p[tile + 1, wv, 0] if tile + 1 < 8 else s[4]
            print(f"  tile {tile} wave {wv}: DMA issue {s[1] - s[0]:6d}  MFMA loop {s[2] - s[1]:6d}  vmcnt wait {s[3] - s[2]:6d}  epilogue {s[4] - s[3]:6d}  to next top (barrier) {nxt - s[4]:6d}")
