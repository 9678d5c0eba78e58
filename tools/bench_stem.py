"""Stand-alone timing of the fused stem + pool front at B = 64 (hot: re-launched on the same buffers; behind a 512-MB fill with FLUSH=1):
    python tools/bench_stem.py         RDPN6D_STEM_V1=1 / RDPN6D_STEM_SKEW=<n> select the kernel variants (read once per process)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr, pack_stem_h2_weight
lib = _lib.load(); dev = torch.device("cuda:0")
B, R = int(os.environ.get("B", 64)), 256
x = torch.rand(B, 6, R, R, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) / 12
wh, inv = pack_stem_h2_weight(w)
sc = (torch.rand(64, device=dev) + 0.5) * inv; sh = torch.randn(64, device=dev) * 0.3
y = torch.empty(B * 64 * 64, 2, 2, 32, dtype=torch.float16, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if os.environ.get("FLUSH") else None
run = lambda: _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(x), B, 6, R, _ptr(wh), _ptr(sc), _ptr(sh), _ptr(y), _ptr(flag), st))
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for _ in range(40):
    if flush is not None: flush.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"stem+pool B={B} v1={os.environ.get('RDPN6D_STEM_V1', '')} skew={os.environ.get('RDPN6D_STEM_SKEW', '0')} flush={bool(flush is not None)}: "
      f"median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us")
