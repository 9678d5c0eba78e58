"""Training-target and pose-error kernels: numpy oracle vs golden vectors from the
reference's own functions (CPU), HIP kernels vs the same golden vectors (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import targets_eval_oracle as orc
from tests.targets_eval_cases import pose_case, target_case

CASES = ((0, 32), (1, 32), (2, 8), (3, 64))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "targets_eval_golden.npz"))


def test_oracle_targets_match_reference(gold):
    for seed, K in CASES:
        xyz, fps, R, ext = target_case(seed, K)
        roi, reg = orc.region_targets(xyz, fps, R, ext)
        assert np.array_equal(reg, gold[f"tgt{seed}_region"])
        assert np.abs(roi - gold[f"tgt{seed}_xyz"]).max() < 1e-6


def test_oracle_pose_errors_match_reference(gold):
    for seed in range(4):
        Re, te, Rg, tg, pts = pose_case(seed)
        got = np.array([orc.add(Re, te, Rg, tg, pts), orc.adi(Re, te, Rg, tg, pts), orc.re(Re, Rg), orc.te(te, tg)])
        assert np.abs(got - gold[f"pose{seed}"]).max() < 1e-12


@pytest.mark.gpu
def test_hip_targets_match_reference(gold):
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    for K in (32, 8, 64):
        seeds = [s for s, k in CASES if k == K]
        cs = [target_case(s, K) for s in seeds]
        roi, reg = ops.region_targets(torch.from_numpy(np.stack([c[0] for c in cs])).to(dev), torch.from_numpy(np.stack([c[1] for c in cs])).to(dev),
                                      torch.from_numpy(np.stack([c[2] for c in cs])).to(dev), torch.from_numpy(np.stack([c[3] for c in cs])).to(dev))
        torch.cuda.synchronize()
        for i, s in enumerate(seeds):
            assert np.array_equal(reg[i].cpu().numpy(), gold[f"tgt{s}_region"].astype(np.int64)), s   # region labels bit-exact
            assert np.abs(roi[i].cpu().numpy() - gold[f"tgt{s}_xyz"]).max() < 2e-7, s                   # <= 1 float32 ulp at O(1)


@pytest.mark.gpu
def test_hip_pose_errors_match_reference(gold):
    from rdpn6d_amd import ops

    dev = torch.device("cuda:0")
    cs = [pose_case(s) for s in range(4)]
    t = lambda i: torch.from_numpy(np.stack([c[i] for c in cs])).to(dev)  # noqa: E731
    out = ops.pose_errors(t(0), t(1), t(2), t(3), t(4))  # per-pose point sets
    torch.cuda.synchronize()
    for s in range(4):
        assert np.abs(out[s].cpu().numpy() - gold[f"pose{s}"]).max() < 1e-10, (s, out[s].cpu().numpy(), gold[f"pose{s}"])
    # shared point set + a big one that spills out of LDS
    big = torch.from_numpy((np.random.default_rng(0).random((7000, 3)) - 0.5) * 0.2).to(dev)
    o2 = ops.pose_errors(t(0)[:2], t(1)[:2], t(2)[:2], t(3)[:2], big)
    ref = [orc.add(cs[i][0], cs[i][1], cs[i][2], cs[i][3], big.cpu().numpy()) for i in range(2)]
    assert np.abs(o2[:, 0].cpu().numpy() - np.array(ref)).max() < 1e-10
    refi = orc.adi(cs[0][0], cs[0][1], cs[0][2], cs[0][3], big.cpu().numpy())
    assert abs(o2[0, 1].item() - refi) < 1e-10
