"""row A8 on the GPU: rdpn6d_select_correspondences_f32 against the golden vectors of the reference's own
get_out_coor / get_out_mask / get_img_model_points_with_coords2d (tests/golden/select_golden.npz) - BIT-EXACT point lists,
counts, selection masks and normalised masks; called through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_select_correspondences_bit_exact_vs_reference_golden(golden_dir):
    from oracle import select_oracle
    from rdpn6d_amd import ops
    from tests.select_cases import IM_H, IM_W, select_case

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "select_golden.npz"))
    for seed, thr in ((0, 0.5), (1, 0.5), (2, 0.3)):
        c = select_case(seed)
        B = c["mask"].shape[0]
        maps = np.concatenate([c["mask"], c["coor_x"], c["coor_y"], c["coor_z"], np.zeros((B, 33, 64, 64), np.float32)], 1)  # (B,37,64,64)
        # RDPN layout of roi_coord_2d: [depth_x, depth_y, depth_z, u, v] - the 2D coordinates are the last two channels
        c5 = np.concatenate([np.full((B, 3, 64, 64), 7.0, np.float32), c["coord2d"]], 1)
        ip, mp, cnt, sel, nm = ops.select_correspondences(torch.from_numpy(maps).to(dev), torch.from_numpy(c5).to(dev),
                                                          torch.from_numpy(c["extent"]).to(dev), IM_H, IM_W, mask_thr=thr, return_masks=True)
        torch.cuda.synchronize()
        ip, mp, cnt, sel, nm = ip.cpu().numpy(), mp.cpu().numpy(), cnt.cpu().numpy(), sel.cpu().numpy(), nm.cpu().numpy()
        assert np.array_equal(nm, gold[f"s{seed}_out_mask"][:, 0], equal_nan=True)
        for b in range(B):
            gi, gm = gold[f"s{seed}_b{b}_image_points"], gold[f"s{seed}_b{b}_model_points"]
            assert cnt[b] == len(gi) == sel[b].sum(), (seed, b, cnt[b], len(gi))
            assert np.array_equal(ip[b, :cnt[b]], gi) and np.array_equal(mp[b, :cnt[b]], gm), (seed, b)
            _, _, osel = select_oracle.select_correspondences(gold[f"s{seed}_out_mask"][b, 0], np.stack([c["coor_x"][b, 0], c["coor_y"][b, 0], c["coor_z"][b, 0]], -1),
                                                              c["coord2d"][b].transpose(1, 2, 0), IM_H, IM_W, c["extent"][b], thr)
            assert np.array_equal(sel[b].astype(bool), osel)
        # the reference call site's channels "as given" (0 / 1) are selectable too
        ip01, _, cnt01 = ops.select_correspondences(torch.from_numpy(maps).to(dev), torch.from_numpy(c5).to(dev), torch.from_numpy(c["extent"]).to(dev),
                                                    IM_H, IM_W, mask_thr=thr, u_ch=0, v_ch=1)
        assert torch.equal(cnt01.cpu(), torch.from_numpy(cnt)) and float(ip01[0, 0, 0]) == 7.0 * IM_W


def test_select_correspondences_on_model_output_and_full_batch():
    """B = 64 crops of model-shaped maps (37 channels): per-crop results equal the numpy restatement, whatever the batch slot."""
    from oracle import select_oracle
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    B = 64
    maps = rng.standard_normal((B, 37, 64, 64)).astype(np.float32)
    maps[:, 1:4] = rng.random((B, 3, 64, 64), dtype=np.float32)
    c5 = rng.random((B, 5, 64, 64), dtype=np.float32)
    ext = (rng.random((B, 3), dtype=np.float32) * 0.2 + 0.05).astype(np.float32)
    ip, mp, cnt = ops.select_correspondences(torch.from_numpy(maps).to(dev), torch.from_numpy(c5).to(dev), torch.from_numpy(ext).to(dev), 480, 640)
    torch.cuda.synchronize()
    nm = select_oracle.out_mask_l1(maps[:, :1])
    for b in range(B):
        oi, om, _ = select_oracle.select_correspondences(nm[b, 0], maps[b, 1:4].transpose(1, 2, 0), c5[b, 3:5].transpose(1, 2, 0), 480, 640, ext[b])
        n = int(cnt[b])
        assert n == len(oi) and np.array_equal(ip[b, :n].cpu().numpy(), oi) and np.array_equal(mp[b, :n].cpu().numpy(), om), b
