"""BUILD-CONTAINER ONLY: golden vectors of the symmetric point-matching loss from the reference's OWN PyPMLoss
(core/gdrn_modeling/losses/pm_loss.py) and get_closest_rot_batch (core/utils/pose_utils.py), imported from
/root/reference with stub third-party packages.

  python tools/oracle/gen_pm_sym_golden.py     # writes tests/golden/pm_sym_golden.npz
Stored: the chosen targets, loss_PM_R (symmetric and plain) and d(loss_PM_R)/d(pred_rots) for the seeded case of
tests/pm_sym_cases.py (inputs are regenerated from the seed, their SHA-256 is recorded)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))
import ref_stubs  # noqa: E402

ref_stubs.install()
from core.gdrn_modeling.losses.pm_loss import PyPMLoss  # noqa: E402
from core.utils.pose_utils import get_closest_rot_batch  # noqa: E402
from rdpn6d_amd import synth  # noqa: E402
from tests.pm_sym_cases import make_case  # noqa: E402


def main():
    c = make_case()
    pred = torch.from_numpy(c["pred_rots"]).requires_grad_(True)
    gt, pts, ext = (torch.from_numpy(c[k]) for k in ("gt_rots", "points", "extents"))
    out = {"sha256_inputs": synth.sha256_of([c[k] for k in ("pred_rots", "gt_rots", "points", "extents")])}
    closest = get_closest_rot_batch(pred, gt, sym_infos=c["sym_infos"])
    out["closest_gt_rots"] = closest.numpy()
    print("targets changed by symmetry:", int((closest - gt).abs().amax(dim=(1, 2)).gt(1e-6).sum()), "of", gt.shape[0])
    for name, sym in (("sym", True), ("plain", False)):
        fn = PyPMLoss(loss_type="L1", beta=1.0, reduction="mean", loss_weight=1.0, norm_by_extent=True, symmetric=sym,
                      disentangle_t=False, disentangle_z=True, t_loss_use_points=True, r_only=True)  # GDRN.py:490-501
        ld = fn(pred_rots=pred, gt_rots=gt, points=pts, pred_transes=None, gt_transes=None, extents=ext,
                sym_infos=c["sym_infos"])
        assert list(ld) == ["loss_PM_R"]
        (g,) = torch.autograd.grad(ld["loss_PM_R"], pred)
        out[f"loss_PM_R_{name}"] = np.float64(ld["loss_PM_R"].item())
        out[f"grad_pred_rots_{name}"] = g.numpy()
        print(name, "loss_PM_R", ld["loss_PM_R"].item())
    path = os.path.join(ROOT, "tests", "golden", "pm_sym_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
