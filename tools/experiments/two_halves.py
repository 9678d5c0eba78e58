"""Experiment (round 4): does the step run faster as TWO half-batch pipelines on two HIP streams than as one batch-64 pipeline?

Every trunk launch of the batch-64 plan is one round of 256 workgroups that all load, compute and store in phase (prologue / epilogue
HBM bursts with the matrix pipe idle: profiles/r3_probe_tile_kernel.md) and the step's tail is a chain of launch-bound small kernels.
Two independent half-batch chains in flight overlap one chain's bursts and small kernels with the other's K loops.

    python tools/experiments/two_halves.py [steps]
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from rdpn6d_amd import _lib, synth  # noqa: E402
from rdpn6d_amd.gdrn import InferencePlan  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev)
model.cfg.TEST.USE_PNP = False
lib = _lib.load()
t = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(64, seed=100).items()}
names = ("roi_img", "roi_coord_2d", "fps", "roi_cam", "roi_center", "roi_wh", "resize_ratio")


def args_of(sl):
    return [t[k][sl].float().contiguous() for k in names]


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# --- one batch-64 pipeline
full = InferencePlan(model, 64, dev)
full.bind_outputs(fresh=False)
a64 = args_of(slice(0, 64))
ms_full = timed(lambda: full.run(*a64), steps)
ref = {k: getattr(full, k).clone() for k in ("rot", "trans")}
ref_maps = full.out_nchw.clone()
print(f"one batch-64 pipeline: {ms_full:.3f} ms / step = {64 / ms_full * 1e3:.0f} crops/s")

# --- two batch-32 pipelines on two streams
for share in (1,):  # (a build whose kernel selection sized every launch for half of the CUs - "share 2" - was slower still: 8.6 ms)
    halves = [InferencePlan(model, 32, dev) for _ in range(2)]
    for h in halves:
        h.bind_outputs(fresh=False)
    a32 = [args_of(slice(0, 32)), args_of(slice(32, 64))]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def both():
        cur = torch.cuda.current_stream()
        for h, a, s in zip(halves, a32, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                h.run(*a)
        for s in streams:
            cur.wait_stream(s)

    def both_free():  # no join between steps: the two chains drift apart freely
        for h, a, s in zip(halves, a32, streams):
            with torch.cuda.stream(s):
                h.run(*a)

    def seq():
        for h, a in zip(halves, a32):
            h.run(*a)

    ms_seq = timed(seq, steps)
    ms_two = timed(both, steps)
    ms_free = timed(both_free, steps)
    torch.cuda.synchronize()
    rot = torch.cat([h.rot for h in halves])
    maps = torch.cat([h.out_nchw for h in halves])
    print(f"chip share {share}: two batch-32 pipelines, one stream: {ms_seq:.3f} ms | two streams joined per step: {ms_two:.3f} ms = "
          f"{64 / ms_two * 1e3:.0f} crops/s | two streams free-running: {ms_free:.3f} ms = {64 / ms_free * 1e3:.0f} crops/s | "
          f"max |maps - batch-64 maps| {float((maps - ref_maps).abs().max()):.2e}, rot {float((rot - ref['rot']).abs().max()):.2e}")
    del halves
    torch.cuda.empty_cache()
