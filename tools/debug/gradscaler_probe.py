"""Root cause of the skipped GradScaler steps in tests/test_gpu_fp16.py::test_fp16_training_through_the_reference_loop_with_gradscaler
(VERDICT r3 item 5b / ADVICE r3): which tensor is non-finite, at which loss scale, in which step - per step: the scale, the loss, the
parameter gradients that are not finite (in backward order), and for every fp16 activation-gradient buffer its largest finite
magnitude and its inf / nan count.    python tools/debug/gradscaler_probe.py [lr] [steps] [init_scale]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rdpn6d_amd import synth  # noqa: E402
from rdpn6d_amd.config import gdrn_base_cfg  # noqa: E402
from rdpn6d_amd.gdrn import build_model_optimizer  # noqa: E402


def main():
    lr = float(sys.argv[1]) if len(sys.argv) > 1 else 2e-3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    init = float(sys.argv[3]) if len(sys.argv) > 3 else 65536.0
    dev = torch.device("cuda:0")
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    cfg.SOLVER.AMP.ENABLED, cfg.SOLVER.AMP.DTYPE = True, "fp16"
    cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=lr, weight_decay=0)
    model, opt = build_model_optimizer(cfg)
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    inp = synth.make_inputs(4, seed=0)
    b = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(4, inp)}.items()}
    scaler = torch.amp.GradScaler("cuda", init_scale=init)
    print(f"lr {lr} init_scale {init} fuse_stats {os.environ.get('RDPN6D_BN_FUSE_STATS', '1')} fuse_bwd {os.environ.get('RDPN6D_BN_FUSE_BWD', '1')} "
          f"mfma_stem {os.environ.get('RDPN6D_MFMA_STEM', '1')}")
    names = [n for n, _ in model.named_parameters()]
    for it in range(steps):
        _, ld = model(b["roi_img"], gt_xyz=b["roi_xyz"], gt_mask_trunc=b["roi_mask_trunc"], gt_mask_visib=b["roi_mask_visib"],
                      gt_mask_obj=b["roi_mask_obj"], gt_region=b["roi_region"], gt_ego_rot=b["ego_rot"], gt_points=b["roi_points"],
                      sym_infos=None, gt_trans=b["trans"], gt_trans_ratio=b["roi_trans_ratio"], roi_classes=b["roi_cls"],
                      roi_coord_2d=b["roi_coord_2d"], roi_cams=b["roi_cam"], roi_centers=b["roi_center"], roi_whs=b["roi_wh"],
                      roi_extents=b["roi_extent"], resize_ratios=b["resize_ratio"], do_loss=True, fps=b["fps"])
        losses = sum(ld.values())
        opt.zero_grad(set_to_none=True)
        scale = scaler.get_scale()
        scaler.scale(losses).backward()
        torch.cuda.synchronize()
        eng = model.train_engine(4, dev)
        bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
        print(f"step {it}: scale {scale:.0f} loss {losses.item():.4f} losses " + " ".join(f"{k[5:]}={v.item():.3f}" for k, v in ld.items())
              + f" | non-finite parameter gradients: {len(bad)} / {len(names)}" + (f" (last in backward order: {bad[0]}; first: {bad[-1]})" if bad else ""))
        rows = []
        for k, v in eng.bufs.items():
            if k.startswith(("d:", "dres:")) and v.dtype == torch.float16:
                f = v.float()
                fin = torch.isfinite(f)
                rows.append((k, float(f[fin].abs().max()) if fin.any() else float("nan"), int((~fin).sum()), v.numel()))
        hot = [r for r in rows if r[2] > 0 or r[1] > 6000]
        for k, mx, nbad, n in sorted(hot, key=lambda r: -r[2])[:12]:
            print(f"     {k:28s} max finite |g| {mx:9.1f}  non-finite {nbad} / {n}")
        if not hot:
            top = sorted(rows, key=lambda r: -r[1])[:3]
            print("     largest fp16 activation gradients: " + ", ".join(f"{k} {mx:.1f}" for k, mx, _, _ in top))
        # forward activations close to the fp16 limit?
        acts = [(k, float(v.float().abs().max())) for k, v in eng.bufs.items() if k.startswith(("raw:", "act:")) and v.dtype == torch.float16]
        k, mx = max(acts, key=lambda r: r[1])
        print(f"     largest fp16 forward value: {k} {mx:.1f}")
        scaler.step(opt)
        scaler.update()


if __name__ == "__main__":
    main()
