"""BASELINE configurations C4 and C3 exercised at their per-GPU workload (SURVEY.md section 8a sizes table):

  C4  YCB-V 21-object inference: B = 64 per GPU, YCB-V camera, MASK_ATTENTION = "mul", one RANSAC instance per crop with the
      outlier ratio swept 0 - 70 %;
  C3  LM-O training in bf16 (mixed precision): B = 32 per GPU.

Full-size runs are checked through size-independent properties (every copy of a crop gives the same bits; the distinct
crops meet the bounds of the small case) plus direct comparison with the oracle where it finishes in seconds."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_c4_ycbv_b64_mask_attention_mul_vs_oracle_and_copies(golden_dir):
    """C4 forward: 64 crops = 16 shuffled copies of 4 distinct YCB-V-camera crops, MASK_ATTENTION = mul, TEST.USE_PNP on.
    The distinct crops are compared with the (reference-pinned) torch-CPU oracle at the bare tolerances of the well-conditioned
    fixture (maps 1e-4, pose 1e-4 whenever the arg-max maps agree); all copies must give identical bits (maps, pose, RANSAC)."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    dev = torch.device("cuda:0")
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda", num_classes=21)
    cfg.TEST.USE_PNP, cfg.TEST.PNP_TYPE = True, "ransac_kabsch"
    model, _ = build_model_optimizer(cfg)
    orc = model_oracle.GDRNOracle(32, "mul")
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in orc.state_dict().items()}, seed=1234)
    orc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    inp = synth.make_inputs(4, seed=21, cam="ycbv")
    assert abs(float(inp["roi_cam"][0, 0, 0]) - 1066.778) < 1e-3 and inp["roi_cls"].max() < 21
    tc = {k: torch.from_numpy(v) for k, v in inp.items()}
    model_oracle.calibrate_bn(orc, tc["roi_img"])
    model.load_state_dict(orc.state_dict(), strict=True)
    model.eval()
    args = lambda d: (d["roi_img"], d["roi_coord_2d"], d["fps"], d["roi_cam"], d["roi_center"], d["roi_wh"], d["resize_ratio"])  # noqa: E731
    with torch.no_grad():
        oo = orc(*args(tc))
    order = np.concatenate([np.arange(4), np.random.default_rng(4).permutation(np.repeat(np.arange(4), 15))])
    idx = torch.from_numpy(order)
    t64 = {k: v[idx].contiguous().to(dev) for k, v in tc.items()}
    with torch.no_grad():
        o = model(t64["roi_img"], roi_classes=t64["roi_cls"], roi_coord_2d=t64["roi_coord_2d"], roi_cams=t64["roi_cam"],
                  roi_centers=t64["roi_center"], roi_whs=t64["roi_wh"], roi_extents=t64["roi_extent"], resize_ratios=t64["resize_ratio"],
                  do_loss=False, fps=t64["fps"])
    torch.cuda.synchronize()
    plan = model.plan(64, dev)
    assert plan.x3_trunk and plan.mask_attention == "mul"
    first = {c: int(np.where(order == c)[0][0]) for c in range(4)}
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region", "rot", "trans"):
        v = o[k].cpu()
        assert torch.isfinite(v.float()).all(), k
        for s in range(64):
            assert torch.equal(v[s], v[first[int(order[s])]]), (k, s)
    # the RANSAC draw is decorrelated per batch slot (the hash takes the crop index), so copies of a crop sample different
    # minimal sets: their results are valid poses / sentinels of their own, not identical bits
    pp, ni = o["pnp_pose"].cpu(), o["pnp_num_inliers"].cpu()
    assert torch.isfinite(pp).all() and (ni >= 0).all() and o["pnp_inlier_mask"].shape == (64, 4096)
    solved = ni >= 3
    Rp = pp[solved][:, :9].reshape(-1, 3, 3).double()
    if len(Rp):
        assert (Rp @ Rp.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max().item() < 1e-5
    assert (pp[~solved] == -100).all()
    am = plan.argmax.cpu().numpy().reshape(64, -1)[:4]
    am_o = oo["region_argmax"].numpy().reshape(4, -1)
    top2 = oo["region"][:, 1:].topk(2, dim=1).values
    near_ties = int(((top2[:, 0] - top2[:, 1]) < 1e-4).sum())
    flips = int((am != am_o).sum())
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        err = (o[k][:4].cpu().double() - oo[k].double()).abs().max().item()
        print(f"C4 {k}: HIP-vs-oracle max-abs {err:.2e}")
        assert err <= 1e-4, (k, err)
    wr = [_rel(o["rot"][i].cpu().numpy().astype(np.float64), oo["rot"][i].numpy().astype(np.float64)) for i in range(4)]
    wt = [_rel(o["trans"][i].cpu().numpy().astype(np.float64), oo["trans"][i].numpy().astype(np.float64)) for i in range(4)]
    print(f"C4: arg-max flips {flips} (pixels whose oracle top-2 gap is < 1e-4: {near_ties}); pose rel err per crop R {wr} t {wt}")
    assert flips <= near_ties
    per_crop_flips = (am != am_o).sum(1)
    for i in range(4):
        tol = 1e-4 if per_crop_flips[i] == 0 else 1e-2  # one flipped pixel moves three ConvPnPNet input channels by O(1)
        assert wr[i] <= tol and wt[i] <= tol, (i, wr[i], wt[i], int(per_crop_flips[i]))
    R = o["rot"].cpu().double()
    assert (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max().item() < 1e-5


def test_c4_ransac_stress_b64_outliers_0_to_70_percent(oracle_lib):
    """C4's per-wavefront RANSAC stress: 64 crops in ONE launch, up to ~1 600 correspondences each, outlier ratio swept 0 .. 70 %
    across the batch (+10 % depth holes, 1 mm noise), 100 hypotheses, confidence 0.99.  Inlier masks / counts / winning hypothesis
    bit-exact vs the C oracle for every crop; the known pose is recovered (< 0.5 deg, < 2 mm) for every crop up to 50 % outliers,
    and wherever the winner holds more than 80 % of the clean correspondences beyond that (at 70 % a clean 3-sample is drawn with
    probability 0.027 per hypothesis: 100 hypotheses find one 93 % of the time - the interface's iteration count, not a defect)."""
    from rdpn6d_amd import ops
    from tests.ransac_cases import make_case, pose_errors
    from tests.test_ransac_oracle import run_oracle

    dev = torch.device("cuda:0")
    B = 64
    ratios = np.linspace(0.0, 0.7, B)
    c = make_case(B=B, K=32, side=64, outliers=ratios, seed=77)
    pose_o, nin_o, msk_o, best_o = run_oracle(oracle_lib, c, seed=5)
    g = {k: torch.from_numpy(c[k]).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
    pose, nin, msk, best = ops.ransac_kabsch(g["out_nchw"], g["coord2d"], g["fps"], g["extents"], g["ratios"], g["argmax"], seed=5)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), best_o) and np.array_equal(nin.cpu().numpy(), nin_o) and np.array_equal(msk.cpu().numpy(), msk_o)
    dp = np.abs(pose.cpu().numpy() - pose_o).max(axis=1)
    for b in np.argsort(-dp)[:4]:
        print(f"C4 RANSAC crop {b}: outliers {ratios[b]:.2f} n_inliers {int(nin[b])} clean {int(c['clean'][b].sum())} |pose - oracle| {dp[b]:.2e}")
    # the refit (fp64 Horn / Jacobi on the winner's inliers) is reproduced to 1e-5 wherever the winner is a real consensus set; a
    # lost crop's "winner" is a handful of accidental inliers whose scatter matrix is near-singular (the eigenvector is then
    # decided by the summation order of the fixed reduction tree vs the oracle's sequential sum)
    real = nin.cpu().numpy() >= 50
    assert np.abs(pose.cpu().numpy() - pose_o)[real].max() < 1e-5
    recovered, solid = 0, 0
    for b in range(B):
        re, te = pose_errors(pose[b].cpu().numpy(), c["R"][b], c["t"][b])
        ok = re < 0.5 and te < 0.002
        recovered += ok
        good_winner = int(nin[b]) > 0.8 * int(c["clean"][b].sum())
        solid += good_winner
        if ratios[b] <= 0.5 or good_winner:
            assert ok, (b, float(ratios[b]), re, te, int(nin[b]), int(c["clean"][b].sum()))
    print(f"C4 RANSAC stress: pose recovered on {recovered} / {B} crops (outliers 0..70 %), winner with > 80 % of the clean points on {solid}")
    assert recovered >= 58


@pytest.mark.parametrize("B", [4, 64, 130])
def test_ransac_split_over_workgroups_is_bit_identical(B):
    """round 4: with fewer crops than CUs the 100 wavefront-hypotheses of a crop are spread over up to four workgroups (global
    scoreboard, second launch for scan + refit: rdpn6d_ransac_kabsch_ws).  Same hypotheses, same counts, same scan order - every
    output must be BIT-identical to the one-workgroup-per-crop form, plain and network-initialised, also where the split is not taken
    (B = 130: more than 128 crops -> one part)."""
    from rdpn6d_amd import ops
    from tests.ransac_cases import make_case

    dev = torch.device("cuda:0")
    c = make_case(B=B, K=32, side=64, outliers=np.linspace(0.0, 0.7, B), seed=31)
    g = {k: torch.from_numpy(c[k]).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
    args = (g["out_nchw"], g["coord2d"], g["fps"], g["extents"], g["ratios"], g["argmax"])
    netp = torch.from_numpy(np.concatenate([c["R"].reshape(B, 9), c["t"].reshape(B, 3)], 1).astype(np.float32)).to(dev)
    for kw in (dict(), dict(iters=20, net_pose=netp, net_mode="ransac"), dict(net_pose=netp, net_mode="iter"), dict(iters=7)):
        a = ops.ransac_kabsch(*args, seed=9, split=True, **kw)
        b = ops.ransac_kabsch(*args, seed=9, split=False, **kw)
        torch.cuda.synchronize()
        for x, y, nm in zip(a, b, ("pose", "n_inliers", "mask", "best_hyp")):
            assert torch.equal(x, y), (B, kw.keys(), nm)


def test_c3_amp_training_at_b32_copies_and_small_batch_losses():
    """C3's per-GPU batch in mixed precision (cfg.SOLVER.AMP.ENABLED, bf16 storage): B = 32 = 8 copies of 4 crops has the same
    batch statistics as those 4 crops alone, so the nine losses must equal the B = 4 AMP losses, every copy of a crop must
    decode to the same pose bits; the gradients of this batch size are held by the layer-local float64 recompute test."""
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from rdpn6d_amd.train import TrainEngine

    dev = torch.device("cuda:0")
    inp = synth.make_inputs(4, seed=50)
    gt = synth.make_train_gt(4, inp)
    cfg = gdrn_base_cfg(mask_attention="mul", device="cuda")
    cfg.SOLVER.AMP.ENABLED = True
    model, _ = build_model_optimizer(cfg)
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    res = {}
    keys = ["rot_head_net.features.3.weight", "rot_head_net.features.18.weight", "backbone.layer2.0.conv1.weight", "pnp_net.fc1.weight",
            "backbone.conv1.weight"]
    for B, amp in ((4, True), (32, True)):
        model.cfg.SOLVER.AMP.ENABLED = amp
        model.load_state_dict(sd, strict=True)
        rep = np.tile(np.arange(4), B // 4)
        batch = {k: torch.from_numpy(np.ascontiguousarray(v[rep] if v.shape[0] == 4 else v)).to(dev) for k, v in {**inp, **gt}.items()}
        eng = TrainEngine(model, B, dev, amp=amp)
        assert eng.amp == amp and (len(eng.mirrors) > 80) == amp
        losses = {k: v.item() for k, v in eng.forward_backward(batch).items()}
        torch.cuda.synchronize()
        named = dict(model.named_parameters())
        res[(B, amp)] = (losses, {k: named[k].grad.detach().double().cpu().clone() for k in keys}, eng.rot.cpu().clone(), eng.trans.cpu().clone())
        del eng
        torch.cuda.empty_cache()
    l4, g4, _, _ = res[(4, True)]
    l32, g32, rot, trans = res[(32, True)]
    for k in l4:
        print(f"C3 AMP {k}: B=32 {l32[k]:.6f}  B=4 {l4[k]:.6f}")
        # bf16 storage (one ulp = 4e-3): the dense losses agree to that; the pose-branch losses see the region arg-max of bf16
        # logits, where the two kernel paths (tile kernels at B=4, 8-phase 256x256 tiles at B=32) flip different near-tie pixels
        tol = 5e-2 if k in ("loss_PM_R", "loss_centroid", "loss_z") else 5e-3
        assert abs(l32[k] - l4[k]) <= tol * max(1.0, abs(l4[k])), k
    for s in range(32):
        assert torch.equal(rot[s], rot[s % 4]) and torch.equal(trans[s], trans[s % 4]), s
    # Gradients: the parameter gradients of the B = 32 step are checked where a 16-bit step CAN be checked sharply - layer by layer, every
    # weight / BatchNorm gradient recomputed in float64 from the very 16-bit operands the kernels read, at this batch size
    # (tests/test_gpu_c1w.py::test_amp_step_every_layer_gradient_recomputed_from_the_stored_operands[32-bf16]: 150 tensors, worst 2e-5).
    # (A whole-step comparison with the B = 4 step - "as close as that one is to fp32, cos > 0.5" - stood here until round 4; it could
    # not fail for anything short of a sign error)  Here only: finite.
    for k in keys:
        assert torch.isfinite(g32[k]).all(), k
