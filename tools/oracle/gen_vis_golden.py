"""BUILD-CONTAINER ONLY: the per-step ``vis/*`` training scalars of the REAL reference (GDRN.py:306-368 -> EventStorage.put_scalars).

  python tools/oracle/gen_vis_golden.py      # writes tests/golden/vis_scalars_golden.npz

The reference's train forward (well-conditioned fixture: trained-like weights, the training batch of model_c1w.npz, both attention
variants) runs with a recording EventStorage (tools/oracle/ref_stubs.py); stored per variant: the 17 values it pushed, and the
quantities it computed them from (its train-mode pose ``rot, trans``, the raw head outputs ``pred_t_``) - obtained by wrapping the
reference's own ``compute_mean_re_te`` - so that the numpy restatement (oracle.model_oracle.train_vis_scalars) is pinned on exactly
the reference's inputs, and the HIP kernel is then checked against the restatement.
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
from gen_model_golden import GOLD, build_reference  # noqa: E402
from tests.c1w_cases import c1w_state_dict  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    B = 4
    bn = np.load(os.path.join(GOLD, "bn_stats_c1w.npz"))
    inp = synth.make_inputs(B, seed=synth.C1W_TRAIN_INPUT_SEED, res=256, num_regions=32, cam="lm")
    gt = synth.make_train_gt(B, inp)
    tin, tgt = {k: torch.from_numpy(v) for k, v in inp.items()}, {k: torch.from_numpy(v) for k, v in gt.items()}
    out = {"train_input_seed": np.int64(synth.C1W_TRAIN_INPUT_SEED)}
    from core.gdrn_modeling.models import GDRN as ref_gdrn
    from detectron2.utils.events import get_event_storage

    for att in ("none", "mul"):
        ref, _ = build_reference(att)
        sd = c1w_state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, bn)
        ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        ref.train()
        seen = {}
        orig = ref_gdrn.compute_mean_re_te

        def spy(pred_trans, pred_rot, gt_trans, gt_rot, orig=orig, seen=seen):
            seen.update(trans=pred_trans.detach().numpy().copy(), rot=pred_rot.detach().numpy().copy())
            return orig(pred_trans, pred_rot, gt_trans, gt_rot)

        ref_gdrn.compute_mean_re_te = spy
        get_event_storage().scalars.clear()
        try:
            ref(tin["roi_img"].clone(), gt_xyz=tgt["roi_xyz"], gt_xyz_bin=None, gt_mask_trunc=tgt["roi_mask_trunc"],
                gt_mask_visib=tgt["roi_mask_visib"], gt_mask_obj=tgt["roi_mask_obj"], gt_region=tgt["roi_region"],
                gt_ego_rot=tgt["ego_rot"], gt_points=tgt["roi_points"], sym_infos=None, gt_trans=tgt["trans"],
                gt_trans_ratio=tgt["roi_trans_ratio"], roi_classes=tin["roi_cls"], roi_coord_2d=tin["roi_coord_2d"].clone(),
                roi_cams=tin["roi_cam"].clone(), roi_centers=tin["roi_center"], roi_whs=tin["roi_wh"], roi_extents=tin["roi_extent"],
                resize_ratios=tin["resize_ratio"], do_loss=True, fps=tin["fps"])
        finally:
            ref_gdrn.compute_mean_re_te = orig
        vis = dict(get_event_storage().scalars)
        names = sorted(k for k in vis if k.startswith("vis/"))
        assert len(names) == 17, names
        for k in names:
            out[f"{att}_{k}"] = np.float64(vis[k])
        out[f"{att}_pred_rot"], out[f"{att}_pred_trans"] = seen["rot"], seen["trans"]
        # pred_t_ = [vis/tx_net, ty_net, tz_net] of crop 0 only; the restatement needs nothing else of it
        print(att, {k: round(float(vis[k]), 6) for k in names})
    out["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(GOLD, "vis_scalars_golden.npz"), **out)
    print("wrote vis_scalars_golden.npz")


if __name__ == "__main__":
    main()
