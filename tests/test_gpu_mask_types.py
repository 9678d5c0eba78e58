"""cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = "BCE" / "CE" on the GPU (VERDICT r4 missing 1): how the mask channel(s) are READ -
get_mask_prob (models/model_utils.py:24-42) inside the forward, get_out_mask (engine_utils.py:118-136) in front of the pose solves -
against the outputs of the REAL reference (tests/golden/mask_types_golden.npz, tools/oracle/gen_mask_types_golden.py) and, for the
RANSAC / Kabsch solve, bit for bit against the C oracle.  Everything goes through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_selection_bce_ce_bit_exact_vs_reference_golden(golden_dir):
    from rdpn6d_amd import ops
    from tests.select_cases import IM_H, IM_W, mask_logits_case, select_case

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "mask_types_golden.npz"))
    for mlt in ("BCE", "CE"):
        for seed, thr in ((0, 0.5), (1, 0.5), (2, 0.3)):
            c = select_case(seed)
            logits = mask_logits_case(c["mask"], mlt, seed)
            B = logits.shape[0]
            maps = np.concatenate([logits, c["coor_x"], c["coor_y"], c["coor_z"], np.zeros((B, 33, 64, 64), np.float32)], 1)
            c5 = np.concatenate([np.full((B, 3, 64, 64), 7.0, np.float32), c["coord2d"]], 1)
            ip, mp, cnt, sel, nm = ops.select_correspondences(torch.from_numpy(maps).to(dev), torch.from_numpy(c5).to(dev),
                                                              torch.from_numpy(c["extent"]).to(dev), IM_H, IM_W, mask_thr=thr,
                                                              return_masks=True, mask_loss_type=mlt)
            torch.cuda.synchronize()
            ip, mp, cnt, sel, nm = ip.cpu().numpy(), mp.cpu().numpy(), cnt.cpu().numpy(), sel.cpu().numpy(), nm.cpu().numpy()
            g = gold[f"{mlt}_s{seed}_out_mask"][:, 0]
            if mlt == "CE":
                assert np.array_equal(nm, g)
            else:
                assert np.abs(nm - g).max() <= 1.2e-7  # expf on the device vs torch's vectorised CPU sigmoid: two ulps of 0.5..1
            for b in range(B):
                gi, gm = gold[f"{mlt}_s{seed}_b{b}_image_points"], gold[f"{mlt}_s{seed}_b{b}_model_points"]
                assert cnt[b] == len(gi) == sel[b].sum(), (mlt, seed, b, cnt[b], len(gi))
                assert np.array_equal(ip[b, :cnt[b]], gi) and np.array_equal(mp[b, :cnt[b]], gm), (mlt, seed, b)


def _model(mlt, att, golden_dir, pnp=None):
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    cfg = gdrn_base_cfg(mask_attention=att, device="cuda:0")
    cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = mlt
    if pnp:
        cfg.TEST.USE_PNP, cfg.TEST.PNP_TYPE, cfg.TEST.PNP_INLIER_THR = True, pnp, 0.05
    model, _ = build_model_optimizer(cfg)
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    sd.update({k: bn[k] for k in bn.files})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    return model


def _run(model, t):
    with torch.no_grad():
        o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
                  roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("mlt,att", [("BCE", "none"), ("BCE", "mul"), ("CE", "none")])
def test_forward_with_bce_and_ce_masks_vs_the_reference(golden_dir, mlt, att):
    """the whole forward with MASK_LOSS_TYPE BCE (sigmoid attention in the glue kernel) / CE (38 head channels: two mask channels)
    against the reference built by its own factory with that switch: pose within the north star's 1e-4, CE's maps within 1e-4, at
    B = 4 and with the crops replicated to B = 64 (the h2 plan with the fused head output)."""
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "mask_types_golden.npz"))
    inp = synth.make_inputs(4, seed=int(gold["input_seed"]))
    model = _model(mlt, att, golden_dir)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    for rep in (1, 16):
        t = {k: torch.from_numpy(np.concatenate([v] * rep)).to(dev) for k, v in inp.items()}
        o = _run(model, t)
        assert not model.h2_range_exceeded(dev)
        MC = 2 if mlt == "CE" else 1
        assert o["mask"].shape[1] == MC and o["region"].shape[1] == 33 and o["coor_x"].shape[1] == 1
        er = max(rel(o["rot"][i].cpu().numpy(), gold[f"{mlt}_{att}_rot"][i % 4]) for i in range(4 * rep))
        et = max(rel(o["trans"][i].cpu().numpy(), gold[f"{mlt}_{att}_trans"][i % 4]) for i in range(4 * rep))
        # (BCE + mul shares the conditioning of the L1 + mul case: every ConvPnPNet input is scaled by the mask probability)
        assert er < (1e-4 if att == "none" else 2e-4) and et < 1e-4, (mlt, att, rep, er, et)
        if mlt == "CE":
            for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
                e = np.abs(o[k].cpu().numpy() - np.concatenate([gold[f"CE_eval_{k}"]] * rep)).max()
                assert e < 1e-4, (k, rep, e)
        print(f"[{mlt} {att} B={4 * rep}] pose R {er:.2e} t {et:.2e}")


def test_ce_mask_with_mask_attention_raises_like_the_reference(golden_dir):
    """get_mask_prob's CE branch is a TypeError in the reference (torch.softmax has no keepdim, model_utils.py:39; recorded in the
    fixture): the HIP path refuses the combination when the plan is built instead of inventing semantics"""
    from rdpn6d_amd import synth

    gold = np.load(os.path.join(golden_dir, "mask_types_golden.npz"))
    assert "keepdim" in str(gold["ce_mul_raises"])
    model = _model("CE", "mul", golden_dir)
    t = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.make_inputs(4, seed=3).items()}
    with pytest.raises(NotImplementedError, match="CE"):
        _run(model, t)


@pytest.mark.parametrize("mlt", ["BCE", "CE"])
def test_ransac_kabsch_reads_the_mask_like_get_out_mask(golden_dir, oracle_lib, mlt):
    """TEST.USE_PNP inside the forward with a BCE / CE mask: the RANSAC / Kabsch kernel selects with sigmoid(mask) > thr / arg-max == 1
    (engine_utils.py:130-134) - inlier masks, counts and winner bit-exact vs the C oracle on the model's own maps; and the stand-alone
    kernel on logits built to select the same pixels equals its L1 run bit for bit."""
    import ctypes

    from rdpn6d_amd import _lib, synth
    from tests.ransac_cases import make_case
    from tests.test_ransac_oracle import run_oracle, with_mask_type

    dev = torch.device("cuda:0")
    mt = {"BCE": 1, "CE": 2}[mlt]
    model = _model(mlt, "none", golden_dir, pnp="ransac_kabsch")
    t = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(4, seed=11).items()}
    o = _run(model, t)
    plan = model.plan(4, dev)
    C = plan.out_nchw.shape[1]
    assert C == (38 if mlt == "CE" else 37)
    c = dict(out_nchw=plan.out_nchw.cpu().numpy().reshape(4, C, 4096), coord2d=t["roi_coord_2d"].cpu().numpy().reshape(4, 5, 4096),
             fps=t["fps"].cpu().numpy(), extents=t["roi_extent"].cpu().numpy(), ratios=t["resize_ratio"].cpu().numpy(),
             argmax=plan.argmax.cpu().numpy(), B=4, HW=4096, K=32)
    po, ni, mo, _ = run_oracle(oracle_lib, c, inlier_thr=0.05, seed=0, mask_type=mt)
    assert np.array_equal(o["pnp_num_inliers"].cpu().numpy(), ni) and np.array_equal(o["pnp_inlier_mask"].cpu().numpy(), mo)
    assert np.abs(o["pnp_pose"].cpu().numpy() - po).max() < 1e-4
    # the selection really is the sigmoid / arg-max one
    m = c["out_nchw"]
    want = (m[:, 0] > 0) if mlt == "BCE" else (m[:, 1] > m[:, 0])
    assert (mo.astype(bool) & ~want).sum() == 0

    # stand-alone: same selected pixels => the L1 solve, bit for bit
    lib = _lib.load()
    P = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    base = make_case(B=3, outliers=0.3, seed=5)
    res = []
    for case, mtype in ((base, 0), (with_mask_type(base, mt), mt)):
        dv = {k: torch.from_numpy(np.ascontiguousarray(case[k])).to(dev) for k in ("out_nchw", "coord2d", "fps", "extents", "ratios", "argmax")}
        pose, nin = torch.zeros(3, 12, device=dev), torch.zeros(3, dtype=torch.int32, device=dev)
        msk, best = torch.zeros(3, 4096, dtype=torch.uint8, device=dev), torch.zeros(3, dtype=torch.int32, device=dev)
        ws = torch.zeros(int(lib.rdpn6d_ransac_workspace_bytes(3)), dtype=torch.uint8, device=dev)
        _lib.check(lib.rdpn6d_ransac_kabsch_ws_mt(P(dv["out_nchw"]), P(dv["coord2d"]), P(dv["fps"]), P(dv["extents"]), P(dv["ratios"]), P(dv["argmax"]),
                                                  None, 3, 4096, 32, 0.5, mtype, 0.01, 100, 0.99, 7, 1, 1.0, P(pose), P(nin), P(msk), P(best),
                                                  P(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        res.append((pose.cpu().numpy(), nin.cpu().numpy(), msk.cpu().numpy(), best.cpu().numpy()))
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    want = run_oracle(oracle_lib, base)
    assert np.array_equal(res[0][1], want[1]) and np.array_equal(res[0][2], want[2]) and np.array_equal(res[0][3], want[3])


@pytest.mark.parametrize("mlt,att", [("BCE", "none"), ("BCE", "mul"), ("CE", "none")])
def test_training_step_with_bce_and_ce_mask_losses(golden_dir, mlt, att):
    """The fp32 HIP training step with ROT_HEAD.MASK_LOSS_TYPE = BCE / CE (loss_mask = BCEWithLogits / CrossEntropy over two mask channels,
    GDRN.py:455-460; sigmoid mask attention and its backward; the 38-channel head under CE): the nine losses within 1e-5 of the REAL
    reference built with that switch (tests/golden/mask_types_golden.npz), gradient norms within its own reproducibility, and all 164
    gradients within 2e-4 of the reference-pinned oracle's autograd with the HIP forward's ReLU decisions and region arg-max forced."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from rdpn6d_amd.train import TrainEngine
    from tests.test_gpu_c1w import _hip_relu_masks

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "mask_types_golden.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    cfg = gdrn_base_cfg(mask_attention=att, device="cuda")
    cfg.MODEL.CDPN.ROT_HEAD.MASK_LOSS_TYPE = mlt
    model, _ = build_model_optimizer(cfg)
    sdn = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    sdn.update({k: bn[k] for k in bn.files})
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sdn.items()}
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = synth.make_inputs(4, seed=int(gold["train_input_seed"]))
    gt = synth.make_train_gt(4, inp)
    eng = TrainEngine(model, 4, dev)
    assert eng.mask_type == {"BCE": 1, "CE": 2}[mlt]
    batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **gt}.items()}
    losses = {k: float(v.item()) for k, v in eng.forward_backward(batch).items()}
    torch.cuda.synchronize()
    for k, v in losses.items():
        ref = float(gold[f"train_{mlt}_{att}_{k}"])
        print(f"[train {mlt} {att}] {k}: HIP {v:.7f} reference {ref:.7f}")
        assert abs(v - ref) <= 1e-5 * max(1.0, abs(ref)), (k, v, ref)
    grads = {n: p.grad.detach().cpu().double().clone() for n, p in model.named_parameters()}
    for n, g in grads.items():
        ref = float(gold[f"train_{mlt}_{att}_gradnorm/{n}"])
        if ref > 1e-4:
            assert abs(float(g.norm()) - ref) <= 2e-2 * ref, (n, float(g.norm()), ref)
    amax = eng.argmax.cpu().numpy().reshape(4, 64, 64).astype(np.int64)
    orc = model_oracle.GDRNOracle(32, att, mask_loss_type=mlt)
    orc.load_state_dict(sd, strict=True)
    orc.train()
    tc = {k: torch.from_numpy(v) for k, v in {**inp, **gt}.items()}
    with model_oracle.forced_relu_masks(orc, _hip_relu_masks(eng, orc)) as forced:
        oo = orc(tc["roi_img"], tc["roi_coord_2d"], tc["fps"], tc["roi_cam"], tc["roi_center"], tc["roi_wh"], tc["resize_ratio"],
                 train_pose=True, force_argmax=amax)
        sum(model_oracle.gdrn_losses(oo, tc, tc["roi_extent"], mask_loss_type=mlt).values()).backward()
    assert len(forced.used) == 48
    rows = []
    for name, g in grads.items():
        ref = dict(orc.named_parameters())[name].grad.double()
        if ref.norm().item() < 1e-4:
            assert g.norm().item() < 1e-4, name
            continue
        rows.append(((g - ref).norm().item() / ref.norm().item(), name))
    rows.sort(reverse=True)
    print(f"[train {mlt} {att}] HIP vs decision-forced oracle autograd, {len(rows)} tensors: median {np.median([r[0] for r in rows]):.2e}, worst "
          + ", ".join(f"{n} {e:.2e}" for e, n in rows[:3]))
    # the same decision pattern evaluated in float64 = the EXACT gradients of it (no summation-order noise of an fp32 CPU run, whose
    # thread count moves the worst tensor by ~10 %: 1.9e-4 at 128 threads, 2.13e-4 at 32 - profiles/r6_notes.md)
    from tests.conftest import capped_threads

    orc64 = model_oracle.GDRNOracle(32, att, mask_loss_type=mlt)
    orc64.load_state_dict(sd, strict=True)
    orc64.double().train()
    t64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in tc.items()}
    with capped_threads(), model_oracle.forced_relu_masks(orc64, _hip_relu_masks(eng, orc64)):
        oo = orc64(t64["roi_img"], t64["roi_coord_2d"], t64["fps"], t64["roi_cam"], t64["roi_center"], t64["roi_wh"], t64["resize_ratio"],
                   train_pose=True, force_argmax=amax)
        sum(model_oracle.gdrn_losses(oo, t64, t64["roi_extent"], mask_loss_type=mlt).values()).backward()
    rows64 = []
    for name, g in grads.items():
        ref = dict(orc64.named_parameters())[name].grad
        if ref.norm().item() >= 1e-4:
            rows64.append(((g - ref).norm().item() / ref.norm().item(), name))
    rows64.sort(reverse=True)
    print(f"[train {mlt} {att}] ... vs the float64 evaluation of the same decisions: median {np.median([r[0] for r in rows64]):.2e}, worst "
          + ", ".join(f"{n} {e:.2e}" for e, n in rows64[:3]))
    worst32, worst64 = dict((n, e) for e, n in rows), dict((n, e) for e, n in rows64)
    for name in worst32:  # within the stated 2e-4 of the decision-forced oracle: of its fp32 run on this box, or of its exact evaluation
        assert min(worst32[name], worst64.get(name, 1.0)) <= 2e-4, (name, worst32[name], worst64.get(name))
