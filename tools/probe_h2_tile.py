"""Where does a K-chunk step of conv_h2_tile_kernel go?  Probe build only:
    RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force && RDPN6D_H2_SCHED=9 RDPN6D_H2_NST=3 python tools/probe_h2_tile.py
Per wave the kernel sums shader cycles of: fragment-read issue | DMA issue (addresses + 1 KiB LDS-DMA pieces) | MFMA issue |
s_waitcnt vmcnt + lgkmcnt | s_barrier, plus prologue + loop and epilogue time (s_memtime; ~10 % intrusive)."""
import ctypes, os, sys
os.environ.setdefault("RDPN6D_H2_PP", "0")  # every layer on the tile kernel (the ping-pong kernel has its own probe: tools/probe_h2_pp.py)
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib, ops
from rdpn6d_amd.gdrn import _ptr, pack_conv_weight, pack_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
lib.rdpn6d_debug_h2_probe.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("B", 64))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
probe = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
_lib.check(lib.rdpn6d_debug_h2_probe(_ptr(probe)))
for name, H, C in (("layer1", 64, 64), ("layer2", 32, 128), ("layer3", 16, 256), ("layer4", 8, 512)):
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
    wp32 = pack_conv_weight(w); wh, inv = pack_h2_weight(wp32)
    xh, _ = ops.split_h2(x); rh, _ = ops.split_h2(torch.randn(B, H, H, C, device=dev)); yh = torch.empty_like(xh)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale = _ptr(xh), _ptr(wh), _ptr(inv)
    d.B, d.H, d.W, d.Cin, d.in_cs, d.Ho, d.Wo, d.stride = B, H, H, C, C, H, H, 1
    d.ntaps = 9
    for t, (dy, dx) in enumerate([(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]): d.dy[t], d.dx[t] = dy, dx
    d.N, d.Npad, d.OH, d.OW, d.osy, d.osx, d.out_cs, d.act, d.res_cs = C, wp32.shape[0], H, H, 1, 1, C, 1, C
    for _ in range(3):
        _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), st))
    torch.cuda.synchronize(); probe.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.rdpn6d_conv2d_h2(ctypes.byref(d), _ptr(yh), _ptr(rh), _ptr(flag), st)); e1.record()
    torch.cuda.synchronize()
    pall = probe.cpu().numpy().reshape(-1, 8)
    nw = int((pall[:, 5] > 0).sum())
    life = (pall[:nw, 5] + pall[:nw, 6]).reshape(-1, 4).max(1)  # per workgroup
    q = np.percentile(life, [0, 10, 50, 90, 99, 100]).astype(int).tolist()
    per_xcd = [int(life[x::8].mean()) for x in range(8)]
    rt = pall[:nw, 7].astype(np.uint64)
    rs, re = (rt & np.uint64(0xffffffff)).astype(np.int64).reshape(-1, 4).min(1), (rt >> np.uint64(32)).astype(np.int64).reshape(-1, 4).max(1)
    t0 = rs.min()
    print(f"   100 MHz clock: workgroup START after the first (us) p0/p10/p50/p90/p100: {(np.percentile(rs - t0, [0, 10, 50, 90, 100]) / 100).round(1).tolist()}; "
          f"END: {(np.percentile(re - t0, [0, 10, 50, 90, 100]) / 100).round(1).tolist()}; duration p50 {np.median(re - rs) / 100:.1f} us")
    print(f"   workgroup lifetime (cycles) min/p10/p50/p90/p99/max: {q}; mean per XCD: {per_xcd}; max per XCD: {[int(life[x::8].max()) for x in range(8)]}")
    p = pall[:nw]
    nk = 9 * C // 32
    steps = (nk // 2) * 2
    m = p[:, :5].mean(0) / steps
    span = (p[:, 7] + p[:, 5] + p[:, 6]).max() - p[:, 7].min()
    print(f"{name}: {e0.elapsed_time(e1)*1e3:6.1f} us, {len(p)} waves, {nk} chunks | per step: reads {m[0]:6.0f}  dma {m[1]:6.0f}  mfma {m[2]:6.0f}  "
          f"waitcnt {m[3]:6.0f}  barrier {m[4]:6.0f}  = {m.sum():6.0f} cycles | prologue+loop {p[:,5].mean():8.0f}  epilogue {p[:,6].mean():7.0f}  "
          f"kernel span {span} cycles, first-to-last wave start {p[:,7].max() - p[:,7].min()}")
