"""host time vs device time of one training step (is the step launch-bound?): python tools/debug/train_cpu_time.py [bf16|f32]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from rdpn6d_amd import synth
from rdpn6d_amd.parallel import GradBuckets
from rdpn6d_amd.ranger import Ranger

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
B = 32
model, _ = bench.build_model(dev, "mul")
model.cfg.TEST.USE_PNP = False
model.cfg.SOLVER.AMP.ENABLED = dt != "f32"
model.cfg.SOLVER.AMP.DTYPE = dt if dt != "f32" else "bf16"
eng = model.train_engine(B, dev)
buckets = GradBuckets(model)
order = [p for g in ("pnp_net", "rot_head_net", "backbone") for p in getattr(model, g).parameters()]
opt = Ranger(order, lr=1e-4, flat_grad=buckets.flat)
inp = synth.make_inputs(B, seed=200)
batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(B, inp)}.items()}
parts = {}
def step():
    t = time.perf_counter(); eng.forward_losses(batch); parts["fwd"] = parts.get("fwd", 0) + time.perf_counter() - t
    t = time.perf_counter(); eng.backward(on_group_done=buckets.reduce); parts["bwd"] = parts.get("bwd", 0) + time.perf_counter() - t
    t = time.perf_counter(); buckets.finish(); opt.step(); eng.refresh_weights(); parts["opt"] = parts.get("opt", 0) + time.perf_counter() - t
for _ in range(3): step()
torch.cuda.synchronize(); parts.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
    torch.cuda.synchronize()   # host time of a step with an EMPTY queue in front of it = pure launch cost + device time not overlapped
t_sync = (time.perf_counter() - t0) / N
host = {k: v / N * 1e3 for k, v in parts.items()}
parts.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step()
t_host_only = (time.perf_counter() - t0) / N
torch.cuda.synchronize(); t_async = (time.perf_counter() - t0) / N
print(f"{dt}: step with a sync after each {t_sync*1e3:.2f} ms; host-side launch time per step (sync'd run) {host}; "
      f"back-to-back: host returns after {t_host_only*1e3:.2f} ms/step, device done after {t_async*1e3:.2f} ms/step; launches fwd {len(eng.fwd)} bwd groups {len(eng.bwd)}")
