"""BUILD-CONTAINER ONLY: golden vectors from the reference's own xyz_to_region / data_loader arithmetic and
lib.pysixd.pose_error functions (imported from /root/reference with stub third-party packages)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))
import ref_stubs  # noqa: E402

ref_stubs.install()
from core.utils.data_utils import xyz_to_region  # noqa: E402
from lib.pysixd import pose_error  # noqa: E402
from tests.targets_eval_cases import pose_case, target_case  # noqa: E402

out = {}
for seed, K in ((0, 32), (1, 32), (2, 8), (3, 64)):
    xyz, fps, R, ext = target_case(seed, K)
    roi_region, delta = xyz_to_region(xyz, fps)                      # data_utils.py:229-244
    delta = R.dot(delta.reshape(-1, 3).T).T.reshape((64, 64, 3))     # data_loader.py:886-888
    roi_xyz = delta.transpose(2, 0, 1)                               # :898
    roi_xyz[0] = roi_xyz[0] / ext[0] + 0.5                           # :900-902
    roi_xyz[1] = roi_xyz[1] / ext[1] + 0.5
    roi_xyz[2] = roi_xyz[2] / ext[2] + 0.5
    out[f"tgt{seed}_xyz"] = roi_xyz.astype("float32")                # :942
    out[f"tgt{seed}_region"] = roi_region.astype(np.int32)           # :889-890
for seed in range(4):
    Re, te, Rg, tg, pts = pose_case(seed)
    out[f"pose{seed}"] = np.array([pose_error.add(Re, te, Rg, tg, pts), pose_error.adi(Re, te, Rg, tg, pts),
                                   pose_error.re(Re, Rg), pose_error.te(te, tg)], dtype=np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "targets_eval_golden.npz"), **out)
print("wrote", len(out))
