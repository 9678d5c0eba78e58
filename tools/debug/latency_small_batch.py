"""per-call latency of the inference step at per-image batch sizes: python tools/debug/latency_small_batch.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from rdpn6d_amd import synth
dev = torch.device("cuda:0")
for B in [int(b) for b in os.environ.get("BS", "1,2,4,8,12,15").split(",")]:
    model, _ = bench.build_model(dev, "none")
    t = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(B, seed=1).items()}
    with torch.no_grad():
        for _ in range(5): bench.step(model, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): bench.step(model, t)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    p = model.plan(B, dev)
    print(f"B={B:2d}: {dt*1e3:6.3f} ms/step  {B/dt:7.1f} crops/s  fast={p.fast} trunk_fast={p.x3_trunk} h2_pointwise={p.h2_pointwise} x3_launches={p.x3_launches}")
