import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from rdpn6d_amd import synth
from rdpn6d_amd.config import gdrn_base_cfg
from rdpn6d_amd.gdrn import build_model_optimizer
dev = torch.device("cuda:0")
inp = synth.make_inputs(4, seed=0)
b = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(4, inp)}.items()}
cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
cfg.SOLVER.AMP.ENABLED, cfg.SOLVER.AMP.DTYPE = True, "fp16"
cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=2e-3, weight_decay=0)
model, opt = build_model_optimizer(cfg)
sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
for it in range(14):
    _, ld = model(b["roi_img"], gt_xyz=b["roi_xyz"], gt_mask_trunc=b["roi_mask_trunc"], gt_mask_visib=b["roi_mask_visib"],
                  gt_mask_obj=b["roi_mask_obj"], gt_region=b["roi_region"], gt_ego_rot=b["ego_rot"], gt_points=b["roi_points"],
                  sym_infos=None, gt_trans=b["trans"], gt_trans_ratio=b["roi_trans_ratio"], roi_classes=b["roi_cls"],
                  roi_coord_2d=b["roi_coord_2d"], roi_cams=b["roi_cam"], roi_centers=b["roi_center"], roi_whs=b["roi_wh"],
                  roi_extents=b["roi_extent"], resize_ratios=b["resize_ratio"], do_loss=True, fps=b["fps"])
    losses = sum(ld.values())
    opt.zero_grad(set_to_none=True)
    scale = scaler.get_scale()
    scaler.scale(losses).backward()
    eng = model.train_engine(4, dev)
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
    sites = []
    for k, v in eng.bufs.items():
        if k.startswith(("d:", "dres:")) and v.dtype == torch.float16:
            f = v.float()
            nf = int((~torch.isfinite(f)).sum().item())
            mx = float(f[torch.isfinite(f)].abs().max().item()) if nf < f.numel() else float("nan")
            if nf or mx > 40000:
                sites.append((k, nf, f.numel(), -1 if mx != mx else round(mx)))
    print(it, scale, round(losses.item(), 4), "bad params", len(bad), bad[:3], "| buffers near/over the fp16 limit:", sites[:40])
    # unfused recompute of every residual-form BatchNorm backward from the stored buffers
    import ctypes
    from rdpn6d_amd import _lib
    from rdpn6d_amd.gdrn import _ptr
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rep = []
    for r in eng.records:
        if r.get("kind") != "bn" or r["res"] is None or r["dres"] is None:
            continue
        C, M = r["C"], r["M"]
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx, dr = torch.empty_like(r["x_raw"]), torch.empty_like(r["x_raw"])
        mean, istd = eng.bufs["mean:" + r["name"]], eng.bufs["istd:" + r["name"]]
        scr = torch.empty(512 * 1024 * 2 + 4096, dtype=torch.float64, device=dev)
        _lib.check(lib.rdpn6d_bn_backward_fp16(_ptr(r["x_raw"]), r["cs"], r["co"], _ptr(r["dy"]), r["dy_cs"], r["dy_co"], _ptr(r["y"]), r["ycs"], r["yco"],
                                               _ptr(mean), _ptr(istd), _ptr(r["bn"].weight), _ptr(dg), _ptr(db), _ptr(dx), r["cs"], r["co"], _ptr(dr),
                                               C, 0, M, C, 1, _ptr(scr), st))
        torch.cuda.synchronize()
        gw, gb = r["bn"].weight.grad, r["bn"].bias.grad
        e1 = ((dg - gw).abs().max() / (dg.abs().max() + 1e-30)).item()
        e2 = ((db - gb).abs().max() / (db.abs().max() + 1e-30)).item()
        e3 = (dx.float() - r["dx"].float()).abs().max().item() / (dx.float().abs().max().item() + 1e-30)
        rep.append((r["name"], f"{e1:.1e}", f"{e2:.1e}", f"{e3:.1e}"))
    print("   bn2 recompute (dgamma, dbeta, dx rel err):", rep[:6])
    scaler.step(opt)
    scaler.update()
