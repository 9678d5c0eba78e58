python -m pytest tests/test_gpu_h2.py -q -s -k "stem_pool" 2>&1 | grep -v amdgpu | tail -6
python - <<'PY'
import ctypes, torch, sys
sys.path.insert(0, ".")
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr, pack_stem_h2_weight
lib, dev = _lib.load(), torch.device("cuda:0")
B, R = 64, 256
x = torch.rand(B, 6, R, R, device=dev)
wh, inv = pack_stem_h2_weight(torch.randn(64, 3, 7, 7, device=dev) / 12); sc = inv.clone(); sh = torch.zeros(64, device=dev)
y = torch.empty(B * 64 * 64, 2, 2, 32, dtype=torch.float16, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
f = lambda: _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(x), B, 6, R, _ptr(wh), _ptr(sc), _ptr(sh), _ptr(y), _ptr(flag), None))
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("stem_pool_h2 B=64:", e0.elapsed_time(e1) / 20 * 1e3, "us")
PY
