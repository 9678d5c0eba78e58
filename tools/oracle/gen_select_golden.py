"""BUILD-CONTAINER ONLY: golden vectors of row A8 from the reference's own functions (imported from /root/reference with
stub third-party packages): engine_utils.get_out_coor / get_out_mask (torch) and
GDRN_Evaluator.get_img_model_points_with_coords2d (numpy), called in the order process_pnp_ransac calls them
(gdrn_evaluator.py:325-374).  Writes tests/golden/select_golden.npz.

Note on the 2D coordinates: RDPN's roi_coord_2d has FIVE channels and process_pnp_ransac hands all five to a function
documented for HW2 (its reshape(-1, 2) then mis-pairs the rows - SURVEY.md section 8a note on A8).  The function is
captured here as documented: on an HW2 array."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))
import ref_stubs  # noqa: E402

ref_stubs.install()
import types  # noqa: E402

# the reference's own `ref/__init__.py` imports a module (`delta_full`) that its repository does not contain; the evaluator only
# needs the name `ref` at import time, and a class to derive from in place of detectron2's DatasetEvaluator
sys.modules["ref"] = types.ModuleType("ref")
import detectron2.evaluation  # noqa: E402

detectron2.evaluation.DatasetEvaluator = type("DatasetEvaluator", (), {})
from core.gdrn_modeling.engine_utils import get_out_coor, get_out_mask  # noqa: E402
from core.gdrn_modeling.gdrn_evaluator import GDRN_Evaluator  # noqa: E402
from tests.select_cases import IM_H, IM_W, select_case  # noqa: E402

cfg = ref_stubs.to_attr({"MODEL": {"CDPN": {"ROT_HEAD": {"MASK_LOSS_TYPE": "L1", "XYZ_BIN": 64}}}})
out = {}
for seed, thr in ((0, 0.5), (1, 0.5), (2, 0.3)):
    c = select_case(seed)
    B = c["mask"].shape[0]
    xyz = get_out_coor(cfg, *(torch.from_numpy(c[k]) for k in ("coor_x", "coor_y", "coor_z"))).numpy()
    with np.errstate(all="ignore"):
        m = get_out_mask(cfg, torch.from_numpy(c["mask"])).numpy()
    out[f"s{seed}_out_mask"] = m
    for b in range(B):
        xyz_i = xyz[b].transpose(1, 2, 0).copy()
        c2_i = c["coord2d"][b].transpose(1, 2, 0).copy()
        with np.errstate(all="ignore"):
            ip, mp = GDRN_Evaluator.get_img_model_points_with_coords2d(None, np.squeeze(m[b]), xyz_i, c2_i, im_H=IM_H, im_W=IM_W,
                                                                       extent=c["extent"][b], mask_thr=thr)
        out[f"s{seed}_b{b}_image_points"] = np.ascontiguousarray(ip, dtype=np.float32)
        out[f"s{seed}_b{b}_model_points"] = np.ascontiguousarray(mp, dtype=np.float32)
        assert ip.dtype == np.float32 and mp.dtype == np.float32, (ip.dtype, mp.dtype)
        print(seed, b, "n =", len(ip))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "select_golden.npz"), **out)
print("wrote select_golden.npz")
