"""BUILD-CONTAINER ONLY: golden vectors for cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = "BCE" and "CE" from the REAL reference
(/root/reference, imported with stub third-party packages).  Writes tests/golden/mask_types_golden.npz.

  python tools/oracle/gen_mask_types_golden.py

1. Row A8 with get_out_mask's BCE (sigmoid) and CE (arg-max over two mask channels) branches (engine_utils.py:130-134) followed by
   GDRN_Evaluator.get_img_model_points_with_coords2d, in the order process_pnp_ransac calls them (gdrn_evaluator.py:325-374), on
   the seeded cases of tests/select_cases.py (mask logits = the case's mask shifted / scaled so that both signs occur; CE: a
   second, seeded channel).
3. (below) the training step.
2. The whole model built by the reference's own factory with MASK_LOSS_TYPE = "BCE" (MASK_ATTENTION none and mul: get_mask_prob's
   sigmoid branch, models/model_utils.py:35-37) and "CE" (MASK_ATTENTION none; 38 head channels: two mask channels) on the
   well-conditioned weights / inputs of model_c1w.npz: rot, trans and - CE - the dense maps.  With MASK_ATTENTION = mul the
   reference's CE branch raises (torch.softmax(..., keepdim=True)): recorded as `ce_mul_raises`.
3. The training step of the same three models on model_c1w.npz's training batch: the nine losses (loss_mask = BCEWithLogits /
   CrossEntropy, GDRN.py:455-460) and the gradient norm of every parameter.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_stubs  # noqa: E402

ref_stubs.install()
sys.modules["ref"] = types.ModuleType("ref")  # (the reference's ref/__init__.py imports a module its repository lacks)
import detectron2.evaluation  # noqa: E402

detectron2.evaluation.DatasetEvaluator = type("DatasetEvaluator", (), {})
from rdpn6d_amd.config import Config  # noqa: E402  (loads the reference's mmcv-style config files; mmcv itself is a stub here)

from core.gdrn_modeling.engine_utils import get_out_coor, get_out_mask  # noqa: E402
from core.gdrn_modeling.gdrn_evaluator import GDRN_Evaluator  # noqa: E402
from gen_model_golden import GOLD  # noqa: E402
from gen_model_golden_w import ref_eval  # noqa: E402
from oracle import model_oracle  # noqa: E402
from rdpn6d_amd import synth  # noqa: E402
from tests.select_cases import IM_H, IM_W, mask_logits_case, select_case  # noqa: E402


def build_reference(mask_attention, mask_loss_type):
    from core.gdrn_modeling.models import GDRN as ref_gdrn

    ref_gdrn.build_optimizer_with_params = lambda cfg, params: None  # needs the mmcv registry
    cfg = Config.fromfile(os.path.join(ref_stubs.REF_ROOT, "configs/gdrn/lm/a6_cPnP_lm13.py"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.CDPN.BACKBONE.PRETRAINED = ""
    cfg.MODEL.CDPN.PNP_NET.MASK_ATTENTION = mask_attention
    cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = mask_loss_type
    cfg.SOLVER.BASE_LR = 1e-4
    cfg = ref_stubs.to_attr(cfg)
    model, _ = ref_gdrn.build_model_optimizer(cfg)
    return model, cfg


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    # ---- 1. selection
    for mlt in ("BCE", "CE"):
        cfg = ref_stubs.to_attr({"MODEL": {"CDPN": {"ROT_HEAD": {"MASK_LOSS_TYPE": mlt, "XYZ_BIN": 64}}}})
        for seed, thr in ((0, 0.5), (1, 0.5), (2, 0.3)):
            c = select_case(seed)
            logits = mask_logits_case(c["mask"], mlt, seed)
            B = logits.shape[0]
            xyz = get_out_coor(cfg, *(torch.from_numpy(c[k]) for k in ("coor_x", "coor_y", "coor_z"))).numpy()
            m = get_out_mask(cfg, torch.from_numpy(logits)).numpy()
            out[f"{mlt}_s{seed}_out_mask"] = m.astype(np.float32)
            assert m.shape == (B, 1, 64, 64), m.shape
            for b in range(B):
                ip, mp = GDRN_Evaluator.get_img_model_points_with_coords2d(None, np.squeeze(m[b]), xyz[b].transpose(1, 2, 0).copy(),
                                                                           c["coord2d"][b].transpose(1, 2, 0).copy(), im_H=IM_H, im_W=IM_W,
                                                                           extent=c["extent"][b], mask_thr=thr)
                out[f"{mlt}_s{seed}_b{b}_image_points"] = np.ascontiguousarray(ip, dtype=np.float32)
                out[f"{mlt}_s{seed}_b{b}_model_points"] = np.ascontiguousarray(mp, dtype=np.float32)
                print(mlt, seed, b, "n =", len(ip))

    # ---- 2. the whole model
    B = 4
    inp = synth.make_inputs(B, seed=synth.C1W_INPUT_SEED, res=256, num_regions=32, cam="lm")
    tin = {k: torch.from_numpy(v) for k, v in inp.items()}
    out["input_seed"] = np.int64(synth.C1W_INPUT_SEED)
    bn = np.load(os.path.join(GOLD, "bn_stats_c1w.npz"))
    for mlt in ("BCE", "CE"):
        orc = model_oracle.GDRNOracle(num_regions=32, mask_attention="none", mask_loss_type=mlt)
        shapes = {k: tuple(v.shape) for k, v in orc.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.make_trained_like_state_dict(shapes, seed=1234).items()}
        sd.update({k: torch.from_numpy(bn[k]) for k in bn.files})
        out[f"{mlt}_sha256_weights"] = synth.sha256_of([sd[k].numpy() for k in sorted(sd) if not k.endswith("num_batches_tracked")])
        for att in ("none", "mul"):
            ref, _ = build_reference(att, mlt)
            ref.load_state_dict(sd, strict=True)
            ref.eval()
            try:
                o = ref_eval(ref, tin)
            except TypeError as e:
                assert mlt == "CE" and att == "mul", (mlt, att, e)
                out["ce_mul_raises"] = np.array(str(e))
                print("[CE mul] the reference raises:", e)
                continue
            out[f"{mlt}_{att}_rot"], out[f"{mlt}_{att}_trans"] = o["rot"].numpy(), o["trans"].numpy()
            if att == "none":
                for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
                    out[f"{mlt}_eval_{k}"] = o[k].numpy()
            orc.mask_attention = att
            orc.load_state_dict(sd, strict=True)
            orc.eval()
            with torch.no_grad():
                oo = orc(tin["roi_img"], tin["roi_coord_2d"], tin["fps"], tin["roi_cam"], tin["roi_center"], tin["roi_wh"], tin["resize_ratio"])
            for k in ("rot", "trans", "mask", "coor_x", "region"):
                print(f"[{mlt} {att}] oracle vs reference {k}: max abs diff {(oo[k] - o[k]).abs().max().item():.3e}")
    # ---- 3. the training step (GDRN.py:450-463: BCEWithLogits / CrossEntropy mask loss; sigmoid attention) on model_c1w.npz's training batch
    inp = synth.make_inputs(B, seed=synth.C1W_TRAIN_INPUT_SEED, res=256, num_regions=32, cam="lm")
    tin = {k: torch.from_numpy(v) for k, v in inp.items()}
    gt = synth.make_train_gt(B, inp)
    tgt = {k: torch.from_numpy(v) for k, v in gt.items()}
    out["train_input_seed"] = np.int64(synth.C1W_TRAIN_INPUT_SEED)
    for mlt, atts in (("BCE", ("none", "mul")), ("CE", ("none",))):
        orc = model_oracle.GDRNOracle(num_regions=32, mask_attention="none", mask_loss_type=mlt)
        shapes = {k: tuple(v.shape) for k, v in orc.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.make_trained_like_state_dict(shapes, seed=1234).items()}
        sd.update({k: torch.from_numpy(bn[k]) for k in bn.files})
        for att in atts:
            torch.set_num_threads(8)
            ref, _ = build_reference(att, mlt)
            ref.load_state_dict(sd, strict=True)
            ref.train()
            _, losses = ref(tin["roi_img"].clone(), gt_xyz=tgt["roi_xyz"], gt_xyz_bin=None, gt_mask_trunc=tgt["roi_mask_trunc"],
                            gt_mask_visib=tgt["roi_mask_visib"], gt_mask_obj=tgt["roi_mask_obj"], gt_region=tgt["roi_region"],
                            gt_ego_rot=tgt["ego_rot"], gt_points=tgt["roi_points"], sym_infos=None, gt_trans=tgt["trans"],
                            gt_trans_ratio=tgt["roi_trans_ratio"], roi_classes=tin["roi_cls"], roi_coord_2d=tin["roi_coord_2d"].clone(),
                            roi_cams=tin["roi_cam"].clone(), roi_centers=tin["roi_center"], roi_whs=tin["roi_wh"],
                            roi_extents=tin["roi_extent"], resize_ratios=tin["resize_ratio"], do_loss=True, fps=tin["fps"])
            sum(losses.values()).backward()
            for k, v in losses.items():
                out[f"train_{mlt}_{att}_{k}"] = np.float64(v.item())
            for n, p in ref.named_parameters():
                out[f"train_{mlt}_{att}_gradnorm/{n}"] = np.float64(p.grad.double().norm().item())
            # the oracle restatement on the spot
            orc.mask_attention = att
            orc.load_state_dict(sd, strict=True)
            orc.train()
            oo = orc(tin["roi_img"], tin["roi_coord_2d"], tin["fps"], tin["roi_cam"], tin["roi_center"], tin["roi_wh"], tin["resize_ratio"],
                     train_pose=True)
            ol = model_oracle.gdrn_losses(oo, tgt, tin["roi_extent"], mask_loss_type=mlt)
            print(f"[train {mlt} {att}] " + " ".join(f"{k.replace('loss_', '')} {float(v):.6f}/{float(ol[k]):.6f}" for k, v in losses.items()))
    np.savez_compressed(os.path.join(GOLD, "mask_types_golden.npz"), **out)
    print("wrote mask_types_golden.npz", os.path.getsize(os.path.join(GOLD, "mask_types_golden.npz")) / 1e6, "MB on disk")


if __name__ == "__main__":
    main()
