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


def test_out_mask_bce_and_ce_branches_vs_reference_golden(golden_dir):
    """get_out_mask's other two branches (engine_utils.py:130-134; VERDICT r4 missing 1): BCE = sigmoid, CE = arg-max over two mask
    channels - the restatement against the outputs of the reference's own function, and the selection that follows it
    (gdrn_evaluator.py:89-126) bit for bit.  tools/oracle/gen_mask_types_golden.py made the fixture."""
    from tests.select_cases import mask_logits_case

    gold = np.load(os.path.join(golden_dir, "mask_types_golden.npz"))
    for mlt in ("BCE", "CE"):
        for seed, thr in CASES:
            c = select_case(seed)
            logits = mask_logits_case(c["mask"], mlt, seed)
            m = select_oracle.out_mask(logits, mlt)
            g = gold[f"{mlt}_s{seed}_out_mask"]
            assert m.shape == g.shape == (logits.shape[0], 1, 64, 64)
            if mlt == "CE":
                assert np.array_equal(m, g)
                assert not m[:, 0, 41, 10:20].any()  # exact ties: channel 0 wins (torch.argmax returns the first maximum)
            else:
                assert np.abs(m - g).max() <= 1.2e-7  # torch's vectorised sigmoid vs 1 / (1 + exp(-x)): within two ulps of 0.5..1
                assert (g[:, 0, 40, 10:20] == 0.5).all()  # logit exactly 0: sigmoid exactly 0.5, NOT selected at thr 0.5 (strict >)
            xyz = np.concatenate([c["coor_x"], c["coor_y"], c["coor_z"]], 1)
            for b in range(m.shape[0]):
                # the selection from the reference's OWN mask is bit-exact; from the restated mask too unless a pixel sits within an ulp
                # of the threshold (none does in the fixture)
                for mm in (g, m):
                    ip, mp, _ = select_oracle.select_correspondences(mm[b, 0], xyz[b].transpose(1, 2, 0), c["coord2d"][b].transpose(1, 2, 0),
                                                                     IM_H, IM_W, c["extent"][b], thr)
                    assert np.array_equal(ip, gold[f"{mlt}_s{seed}_b{b}_image_points"]), (mlt, seed, b)
                    assert np.array_equal(mp, gold[f"{mlt}_s{seed}_b{b}_model_points"]), (mlt, seed, b)
