"""BUILD-CONTAINER ONLY: capture golden vectors from the REAL reference (/root/reference).

  python tools/oracle/gen_model_golden.py            # writes tests/golden/model_c1.npz (+ bn stats)

Steps:
  1. seeded inputs (C1: B=4, LM K, K=32) and seeded weights from rdpn6d_amd/synth.py (numpy PCG64);
  2. BN running statistics calibrated once with the torch-CPU oracle (train-mode pass, momentum=None)
     and SHIPPED as a fixture (tests/golden/bn_stats_c1.npz) so every box uses identical stats;
  3. the reference model is built by ITS OWN factory (core.gdrn_modeling.models.GDRN.build_model_optimizer)
     from ITS OWN config files, loaded with that state_dict (strict), and run in eval mode (both
     MASK_ATTENTION variants) and in train mode (losses + gradient norms);
  4. outputs are written as fixtures; the oracle restatement is compared on the spot.
The reference never travels: only this script and the .npz data do.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_stubs  # noqa: E402

ref_stubs.install()

from rdpn6d_amd import synth  # noqa: E402
from rdpn6d_amd.config import Config  # noqa: E402
from oracle import model_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
GRAD_KEYS = ["backbone.conv1.weight", "backbone.layer4.2.conv2.weight", "backbone.spatial_net.conv1.weight",
             "rot_head_net.features.0.weight", "rot_head_net.features.21.weight", "pnp_net.fc1.weight"]


def build_reference(mask_attention):
    from core.gdrn_modeling.models import GDRN as ref_gdrn

    ref_gdrn.build_optimizer_with_params = lambda cfg, params: None  # needs the mmcv registry
    cfg = Config.fromfile(os.path.join(ref_stubs.REF_ROOT, "configs/gdrn/lm/a6_cPnP_lm13.py"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED = ""
    cfg.MODEL.CDPN.PNP_NET.MASK_ATTENTION = mask_attention
    cfg.SOLVER.BASE_LR = 1e-4
    cfg = ref_stubs.to_attr(cfg)
    model, _ = ref_gdrn.build_model_optimizer(cfg)
    return model, cfg


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    B = 4
    inp = synth.make_inputs(B, seed=0, res=256, num_regions=32, cam="lm")
    tin = {k: torch.from_numpy(v) for k, v in inp.items()}

    # --- oracle: seeded weights + BN calibration -> full state_dict (the fixture carries the BN stats)
    orc = model_oracle.GDRNOracle(num_regions=32, mask_attention="none")
    shapes = {k: tuple(v.shape) for k, v in orc.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed=1234).items()}
    orc.load_state_dict(sd, strict=True)
    model_oracle.calibrate_bn(orc, tin["roi_img"])
    full_sd = {k: v.clone() for k, v in orc.state_dict().items()}
    bn_stats = {k: v.numpy() for k, v in full_sd.items() if k.endswith(("running_mean", "running_var"))}
    np.savez_compressed(os.path.join(GOLD, "bn_stats_c1.npz"), **bn_stats)

    out = {}
    out["sha256_inputs"] = synth.sha256_of([inp[k] for k in sorted(inp)])
    out["sha256_weights"] = synth.sha256_of([full_sd[k].numpy() for k in sorted(full_sd) if not k.endswith("num_batches_tracked")])

    for att in ("none", "mul"):
        ref, cfg = build_reference(att)
        missing = ref.load_state_dict(full_sd, strict=True)
        assert list(ref.state_dict().keys()) == list(full_sd.keys()), "state_dict key ORDER differs from reference"
        ref.eval()
        with torch.no_grad():
            o = ref(tin["roi_img"].clone(), roi_classes=tin["roi_cls"], roi_coord_2d=tin["roi_coord_2d"].clone(),
                    roi_cams=tin["roi_cam"].clone(), roi_centers=tin["roi_center"], roi_whs=tin["roi_wh"],
                    roi_extents=tin["roi_extent"], resize_ratios=tin["resize_ratio"], do_loss=False, fps=tin["fps"])
        # intermediate quantities captured with the reference's own sub-modules
        with torch.no_grad():
            feat = ref.backbone(tin["roi_img"])
        pref = f"eval_{att}_"
        out[pref + "rot"] = o["rot"].numpy()
        out[pref + "trans"] = o["trans"].numpy()
        if att == "none":
            out["backbone_feat_sample0_ch0_8"] = feat[0, :8].numpy()
            out["backbone_feat_sum"] = feat.double().sum(dim=(2, 3)).numpy()  # (B,1024)
            for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
                out["eval_" + k] = o[k].numpy()
            reg = o["region"]
            prob = torch.softmax(reg[:, 1:], dim=1)
            out["eval_region_argmax"] = prob.reshape(B, 32, -1).argmax(1).reshape(B, 64, 64).numpy().astype(np.int8)
        # oracle vs reference, on the spot
        orc.mask_attention = att
        with torch.no_grad():
            oo = orc(tin["roi_img"], tin["roi_coord_2d"], tin["fps"], tin["roi_cam"], tin["roi_center"], tin["roi_wh"],
                     tin["resize_ratio"])
        for k in ("rot", "trans", "mask", "coor_x", "region"):
            d = (oo[k] - o[k]).abs().max().item()
            print(f"[{att}] oracle vs reference {k}: max abs diff {d:.3e}")
        out[pref + "pred_rot6d"] = oo["pred_rot6d"].numpy()  # (reference does not expose these; oracle's,
        out[pref + "pred_t_"] = oo["pred_t_"].numpy()        #  validated through rot/trans above)

    # --- training path: losses and gradient norms from the reference (MASK_ATTENTION = none, C1)
    ref, cfg = build_reference("none")
    ref.load_state_dict(full_sd, strict=True)
    ref.train()
    gt = synth.make_train_gt(B, inp)
    tgt = {k: torch.from_numpy(v) for k, v in gt.items()}
    _, losses = ref(tin["roi_img"].clone(), gt_xyz=tgt["roi_xyz"], gt_xyz_bin=None, gt_mask_trunc=tgt["roi_mask_trunc"],
                    gt_mask_visib=tgt["roi_mask_visib"], gt_mask_obj=tgt["roi_mask_obj"], gt_region=tgt["roi_region"],
                    gt_ego_rot=tgt["ego_rot"], gt_points=tgt["roi_points"], sym_infos=None, gt_trans=tgt["trans"],
                    gt_trans_ratio=tgt["roi_trans_ratio"], roi_classes=tin["roi_cls"],
                    roi_coord_2d=tin["roi_coord_2d"].clone(), roi_cams=tin["roi_cam"].clone(),
                    roi_centers=tin["roi_center"], roi_whs=tin["roi_wh"], roi_extents=tin["roi_extent"],
                    resize_ratios=tin["resize_ratio"], do_loss=True, fps=tin["fps"])
    total = sum(losses.values())
    total.backward()
    for k, v in losses.items():
        out["train_" + k] = np.float64(v.item())
        print("train", k, v.item())
    named = dict(ref.named_parameters())
    for k in GRAD_KEYS:
        out["train_gradnorm_" + k] = np.float64(named[k].grad.double().norm().item())
        print("gradnorm", k, out["train_gradnorm_" + k])
    out["train_sha256_gt"] = synth.sha256_of([gt[k] for k in sorted(gt)])
    # oracle train-path check
    orc.mask_attention = "none"
    orc.load_state_dict(full_sd)
    orc.train()
    oo = orc(tin["roi_img"], tin["roi_coord_2d"], tin["fps"], tin["roi_cam"], tin["roi_center"], tin["roi_wh"],
             tin["resize_ratio"], train_pose=True)
    ol = model_oracle.gdrn_losses(oo, tgt, tin["roi_extent"])
    for k in losses:
        print(f"oracle train {k}: {ol[k].item():.6f} vs ref {losses[k].item():.6f}")

    np.savez_compressed(os.path.join(GOLD, "model_c1.npz"), **out)
    print("wrote", os.path.join(GOLD, "model_c1.npz"))


if __name__ == "__main__":
    main()
