"""Where does the mixed-precision HIP step leave the 16-bit-operand oracle (oracle.lowp_storage)?  Stage by stage: every stored
activation of the HIP forward (act:* buffers) against the oracle's activation at the same ReLU call site, decisions forced.
    python tools/debug/amp_stage_diff.py [bf16|fp16] [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import model_oracle  # noqa: E402
from rdpn6d_amd import synth  # noqa: E402
from rdpn6d_amd.config import gdrn_base_cfg  # noqa: E402
from rdpn6d_amd.gdrn import build_model_optimizer  # noqa: E402
from rdpn6d_amd.train import TrainEngine  # noqa: E402
from tests.c1w_cases import c1w_state_dict  # noqa: E402
from tests.test_gpu_c1w import _hip_relu_masks  # noqa: E402


def main():
    lp = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dt = torch.bfloat16 if lp == "bf16" else torch.float16
    S = 1.0 if lp == "bf16" else 4096.0
    dev = torch.device("cuda:0")
    bn = np.load(os.path.join(ROOT, "tests", "golden", "bn_stats_c1w.npz"))
    model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention="mul", device="cuda"))
    sd = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    inp = synth.make_inputs(B, seed=50)
    gt = synth.make_train_gt(B, inp)
    eng = TrainEngine(model, B, dev, amp=lp)
    eng.loss_scale = S
    batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **gt}.items()}
    losses = {k: v.item() for k, v in eng.forward_backward(batch).items()}
    torch.cuda.synchronize()
    orc = model_oracle.GDRNOracle(32, "mul")
    orc.load_state_dict(sd, strict=True)
    orc.train()
    tc = {k: torch.from_numpy(v) for k, v in {**inp, **gt}.items()}
    masks = _hip_relu_masks(eng, orc)
    name_of = {("backbone.relu", 0): "act:stem"}
    for li in range(1, 5):
        for bi in range(len(getattr(orc.backbone, f"layer{li}"))):
            name_of[(f"backbone.layer{li}.{bi}.relu", 0)] = f"act:layer{li}.{bi}.c1"
            name_of[(f"backbone.layer{li}.{bi}.relu", 1)] = f"act:layer{li}.{bi}"
    name_of[("backbone.spatial_net.relu", 1)] = "act:pn.c1"
    name_of[("backbone.spatial_net.relu", 2)] = "act:pn.c2"
    name_of[("rot_head_net.features.2", 0)] = "act:head0"
    for i in range(3, 21, 3):
        name_of[(f"rot_head_net.features.{i + 2}", 0)] = f"act:head{i}"
    for i in (0, 3, 6):
        name_of[(f"pnp_net.features.{i + 2}", 0)] = f"act:pnp{i}"
    name_of[("pnp_net.act", 0)] = "act:fc1"
    name_of[("pnp_net.act", 1)] = "act:fc2"
    with model_oracle.lowp_storage(orc, dt), model_oracle.forced_relu_masks(orc, masks, round_dtype=dt) as forced:
        oo = orc(tc["roi_img"], tc["roi_coord_2d"], tc["fps"], tc["roi_cam"], tc["roi_center"], tc["roi_wh"], tc["resize_ratio"],
                 train_pose=True, force_argmax=eng.argmax.cpu().numpy().reshape(B, 64, 64))
        L = model_oracle.gdrn_losses(oo, tc, tc["roi_extent"])
        (sum(L.values()) * S).backward()
    print(f"{lp} B={B}: stored activations, HIP vs 16-bit-operand oracle (relative Frobenius | max abs | share of elements that differ)")
    for key, bufname in name_of.items():
        a = eng.bufs[bufname].float().cpu()
        a = a.permute(0, 3, 1, 2) if a.dim() == 4 else a
        o = forced.outputs[key].float()
        if a.shape != o.shape:
            print(f"  {bufname:22s} shape {tuple(a.shape)} vs {tuple(o.shape)}")
            continue
        d = (a - o)
        print(f"  {bufname:22s} rel {d.norm().item() / max(o.norm().item(), 1e-30):.2e}  max {d.abs().max().item():.2e}  differ {(d != 0).float().mean().item():.2e}")
    ho = eng.head_out.reshape(B, 4096, -1)[:, :, :37].permute(0, 2, 1).reshape(B, 37, 64, 64).cpu()
    om = torch.cat([oo["mask"], oo["coor_x"], oo["coor_y"], oo["coor_z"], oo["region"]], 1).detach()
    print(f"  head_out               rel {((ho - om).norm() / om.norm()).item():.2e}  max {(ho - om).abs().max().item():.2e}")
    for k in losses:
        print(f"  {k:16s} HIP {losses[k]:.6f} oracle {L[k].item():.6f}")
    rows = []
    for name, p in orc.named_parameters():
        ref = p.grad.double() / S
        g = dict(model.named_parameters())[name].grad.detach().cpu().double()
        if ref.norm().item() < 1e-6:
            continue
        rows.append(((g - ref).norm().item() / ref.norm().item(), name, ref.norm().item()))
    print("gradients in FORWARD order (relative error | reference norm):")
    for e, n, rn in rows:
        print(f"  {n:50s} {e:.2e}  {rn:.2e}")


if __name__ == "__main__":
    main()
