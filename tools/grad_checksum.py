"""sha256 of the losses and of every parameter gradient of seeded training steps: the A/B check for a kernel change that claims bit-identity
(run once per build: RDPN6D_LIB=<other librdpn6d_hip.so> python tools/grad_checksum.py).  usage: python tools/grad_checksum.py [B]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from rdpn6d_amd import synth
from rdpn6d_amd.config import gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer
from rdpn6d_amd.train import TrainEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
inp = synth.make_inputs(B, seed=9)
batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(B, inp)}.items()}
for amp in (None, "bf16", "fp16"):
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    if amp:
        cfg.SOLVER.AMP.ENABLED, cfg.SOLVER.AMP.DTYPE = True, amp
    model, opt = build_model_optimizer(cfg)
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=5)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    eng = TrainEngine(model, B, dev, amp=amp)
    if amp == "fp16":
        eng.loss_scale = 1024.0
    h = hashlib.sha256()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        losses = eng.forward_backward(batch)
        torch.cuda.synchronize()
        for v in losses.values():
            h.update(np.float32(v.item()).tobytes())
        for p in model.parameters():
            h.update(p.grad.detach().cpu().numpy().tobytes())
        opt.step()
        eng.refresh_weights()
    print(amp, B, h.hexdigest()[:24], {k: round(float(v.item()), 6) for k, v in list(losses.items())[:3]})
    del eng, model, opt
    torch.cuda.empty_cache()
