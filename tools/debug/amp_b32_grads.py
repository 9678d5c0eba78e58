"""diagnosis: AMP (bf16) training gradients at B=32 (8 copies of 4 crops) vs B=4, per tensor, with optional forced tiles"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdpn6d_amd import synth, _lib
from rdpn6d_amd.config import gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer
from rdpn6d_amd.train import TrainEngine

dev = torch.device("cuda:0")
lib = _lib.load()
inp = synth.make_inputs(4, seed=50); gt = synth.make_train_gt(4, inp)

def run(B, amp, force=None):
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda"); cfg.SOLVER.AMP.ENABLED = amp
    model, _ = build_model_optimizer(cfg)
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    rep = np.tile(np.arange(4), B // 4)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v[rep] if v.shape[0] == 4 else v)).to(dev) for k, v in {**inp, **gt}.items()}
    if force: lib.rdpn6d_conv_bf16_force_tile(*force)
    eng = TrainEngine(model, B, dev, amp=amp)
    L = eng.forward_backward(batch); torch.cuda.synchronize()
    lib.rdpn6d_conv_bf16_force_tile(0, 0)
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}, {k: v.item() for k, v in L.items()}

g4f, _ = run(4, False)
g4, l4 = run(4, True)
g32, l32 = run(32, True)
g32f, _ = run(32, True, force=(128, 128))
def rel(a, b): return ((a - b).norm() / max(b.norm(), 1e-30)).item()
print(f"{'tensor':45s} amp4-vs-fp32  amp32-vs-fp32  amp32-vs-amp4  amp32(128 tiles)-vs-amp4")
for n in g4:
    if g4f[n].norm() < 1e-4 or not (n.endswith("weight") and g4[n].dim() > 1): continue
    print(f"{n:45s} {rel(g4[n], g4f[n]):.2e}   {rel(g32[n], g4f[n]):.2e}   {rel(g32[n], g4[n]):.2e}   {rel(g32f[n], g4[n]):.2e}")
