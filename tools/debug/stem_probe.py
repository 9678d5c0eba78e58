import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rdpn6d_amd import _lib
from rdpn6d_amd.gdrn import _ptr, pack_stem_h2_weight
lib, dev = _lib.load(), torch.device("cuda:0")
B, R = 64, 256
x = torch.rand(B, 6, R, R, device=dev)
wh, inv = pack_stem_h2_weight(torch.randn(64, 3, 7, 7, device=dev) / 12); sc = inv.clone(); sh = torch.zeros(64, device=dev)
y = torch.empty(B * 64 * 64, 2, 2, 32, dtype=torch.float16, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(5): _lib.check(lib.rdpn6d_stem_pool_h2(_ptr(x), B, 6, R, _ptr(wh), _ptr(sc), _ptr(sh), _ptr(y), _ptr(flag), None))
torch.cuda.synchronize()
