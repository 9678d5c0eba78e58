"""row A8 on the CPU: the numpy restatement (oracle/select_oracle.py) against the golden vectors produced by the reference's
own get_out_coor / get_out_mask / get_img_model_points_with_coords2d (tools/oracle/gen_select_golden.py) - bit for bit."""
import os

import numpy as np

from oracle import select_oracle
from tests.select_cases import IM_H, IM_W, select_case

CASES = ((0, 0.5), (1, 0.5), (2, 0.3))


def test_selection_oracle_bit_exact_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "select_golden.npz"))
    seen_small = seen_empty = 0
    for seed, thr in CASES:
        c = select_case(seed)
        m = select_oracle.out_mask_l1(c["mask"])
        assert np.array_equal(m, gold[f"s{seed}_out_mask"], equal_nan=True)
        xyz = np.concatenate([c["coor_x"], c["coor_y"], c["coor_z"]], 1)
        for b in range(m.shape[0]):
            ip, mp, sel = select_oracle.select_correspondences(m[b, 0], xyz[b].transpose(1, 2, 0), c["coord2d"][b].transpose(1, 2, 0),
                                                               IM_H, IM_W, c["extent"][b], thr)
            assert np.array_equal(ip, gold[f"s{seed}_b{b}_image_points"]) and np.array_equal(mp, gold[f"s{seed}_b{b}_model_points"]), (seed, b)
            assert sel.sum() == len(ip)
            seen_small += 0 < len(ip) < 5
            seen_empty += len(ip) == 0
    assert seen_small >= 3 and seen_empty >= 3  # the n < 4 sentinel case and the constant-mask (0/0) case are in the fixture
