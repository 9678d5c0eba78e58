"""time of the RANSAC/Kabsch kernel vs the number of hypotheses (what share is phase 2?)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rdpn6d_amd import ops
from tests.ransac_cases import make_case
dev = torch.device("cuda:0")
for outl in (0.3, 0.0):
    c = make_case(B=64, outliers=outl, seed=1)
    t = {k: torch.from_numpy(v).to(dev) for k, v in c.items() if hasattr(v, "shape")}
    for iters in (1, 8, 16, 32, 64, 100):
        f = lambda: ops.ransac_kabsch(t["out_nchw"].reshape(64, 37, 64, 64), t["coord2d"], t["fps"], t["extents"], t["ratios"], t["argmax"], iters=iters)
        for _ in range(3): f()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"outliers {outl} iters {iters:3d}: {e0.elapsed_time(e1)/20*1e3:7.1f} us")
