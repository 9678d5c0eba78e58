"""time of the RANSAC/Kabsch kernel vs the number of hypotheses, and - with a probe build (RDPN6D_PROBE=1 python -m rdpn6d_amd.build) -
the 100 MHz timestamps of workgroup 0 at its phase boundaries"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rdpn6d_amd import _lib, ops  # noqa: E402
from tests.ransac_cases import make_case  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(_lib.LIB_PATH)
probe = None
if hasattr(lib, "rdpn6d_debug_ransac_probe"):
    probe = torch.zeros(16, dtype=torch.int64, device=dev)
    lib.rdpn6d_debug_ransac_probe(ctypes.c_void_p(probe.data_ptr()))
for outl in (0.3,):
    c = make_case(B=64, outliers=outl, seed=1)
    t = {k: torch.from_numpy(v).to(dev) for k, v in c.items() if hasattr(v, "shape")}
    for iters in (1, 16, 100):
        f = lambda: ops.ransac_kabsch(t["out_nchw"].reshape(64, 37, 64, 64), t["coord2d"], t["fps"], t["extents"], t["ratios"], t["argmax"], iters=iters)  # noqa: E731
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        line = f"outliers {outl} iters {iters:3d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us"
        if probe is not None:
            p = probe.cpu().tolist()
            names = ["min/max", "compaction", "hypotheses", "scan", "refit sweeps", "Horn"]
            line += "  | workgroup 0: " + ", ".join(f"{names[i]} {(p[i + 1] - p[i]) / 100.0:.1f} us" for i in range(6))
        print(line)
