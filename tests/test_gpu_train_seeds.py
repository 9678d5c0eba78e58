"""Training-step parity against the REAL reference on EIGHT UNSEARCHED input seeds (tests/golden/train_c1w_seeds.npz,
tools/oracle/gen_train_golden_seeds.py): model_c1w.npz's training batch was searched for the absence of region arg-max ties; these
eight were not looked at.  Per seed and MASK_ATTENTION variant the fp32 HIP step (TrainEngine, B = 4, batch statistics) must give

  * the six dense losses (mask, the three coordinate losses, region, region_my) within the bare 1e-5 of the reference's - they are continuous in the
    maps, no tie rule applies;
  * the train-mode region arg-max of the reference everywhere outside the recorded tie set (top-2 logit gap < 2e-4, or a pixel the
    reference flips against itself between 1 and 8 threads);
  * the pose-branch losses (PM_R, centroid, z) within 1e-5 on every batch whose arg-max it reproduces exactly - ConvPnPNet sees the maps
    through that decision - and, where a tie pixel took the other region, within the reference's sensitivity to that one pixel (1e-3);
  * every parameter's gradient NORM within 2e-2 of the reference's (the reference's own 1-vs-8-thread norms differ by up to ~1e-3: a
    ReLU network's fp32 backward is reproducible only up to the units at round-off of zero; the sharp, decision-forced gradient test is
    tests/test_gpu_c1w.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FLIPPED_SEEDS = {2}  # seeds whose train-mode arg-max differs from the reference's at a pixel of the recorded tie set (round-5 code)
POSE = ("loss_PM_R", "loss_centroid", "loss_z")  # see the maps through the region arg-max; the other six are continuous in the maps


@pytest.fixture(scope="module")
def seeds_steps(golden_dir):
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from rdpn6d_amd.train import TrainEngine
    from tests.c1w_cases import SEEDS, c1w_state_dict

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "train_c1w_seeds.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    assert tuple(int(s) for s in gold["seeds"]) == SEEDS
    res = {}
    for att in ("none", "mul"):
        model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention=att, device="cuda"))
        sdn = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
        assert synth.sha256_of([sdn[k] for k in sorted(sdn) if not k.endswith("num_batches_tracked")]) == str(gold["sha256_weights"])
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sdn.items()}
        for s in SEEDS:
            model.load_state_dict(sd, strict=True)  # (the forward updates the running statistics)
            model.train()
            inp = synth.make_inputs(4, seed=s)
            gt = synth.make_train_gt(4, inp)
            assert synth.sha256_of([inp[k] for k in sorted(inp)]) == str(gold[f"s{s}_sha256_inputs"])
            assert synth.sha256_of([gt[k] for k in sorted(gt)]) == str(gold[f"s{s}_sha256_gt"])
            eng = TrainEngine(model, 4, dev)
            batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **gt}.items()}
            losses = {k: float(v.item()) for k, v in eng.forward_backward(batch).items()}
            torch.cuda.synchronize()
            res[(att, s)] = (losses, eng.argmax.cpu().numpy().reshape(4, 64, 64).astype(np.int64),
                             {n: float(p.grad.double().norm().item()) for n, p in model.named_parameters()})
            del eng
        del model
        torch.cuda.empty_cache()
    return res, gold


@pytest.mark.parametrize("att", ["none", "mul"])
def test_training_losses_on_eight_unsearched_seeds(seeds_steps, att):
    from tests.c1w_cases import SEEDS

    res, gold = seeds_steps
    tie_gap = float(gold["tie_gap"])
    exact, worst_dense, worst_pose, flipped = 0, 0.0, 0.0, set()
    for s in SEEDS:
        losses, amax, _ = res[(att, s)]
        assert len(losses) == 9
        ref_am = gold[f"s{s}_argmax"].astype(np.int64)
        tie = (gold[f"s{s}_top2_gap"] < tie_gap) | np.unpackbits(gold[f"s{s}_flip_1v8"])[: 4 * 4096].reshape(4, 64, 64).astype(bool)
        diff = amax != ref_am
        assert int((diff & ~tie).sum()) == 0, (s, int((diff & ~tie).sum()))
        flips = int(diff.sum())
        exact += flips == 0
        if flips:
            flipped.add(s)
        row = []
        for k, v in losses.items():
            ref = float(gold[f"s{s}_{att}_{k}"])
            e = abs(v - ref) / max(1.0, abs(ref))
            row.append(f"{k.replace('loss_', '')} {e:.1e}")
            if k not in POSE:
                worst_dense = max(worst_dense, e)
                assert e <= 1e-5, (s, k, v, ref)
            elif flips == 0:
                worst_pose = max(worst_pose, e)
                assert e <= 1e-5, (s, k, v, ref)
            else:
                assert e <= 1e-3, (s, k, v, ref, flips)
        print(f"[train seeds {att}] seed {s}: tie set {int(tie.sum())} px, arg-max differs at {flips} tie px | " + " ".join(row))
    print(f"[train seeds {att}] dense losses worst {worst_dense:.1e}; pose-branch losses worst {worst_pose:.1e} on the {exact} of 8 batches "
          f"whose arg-max is reproduced exactly")
    # the batches on which a tie pixel takes the other region are the RECORDED ones (seed 2: one pixel of its 8-pixel tie set, both
    # attention variants; deterministic kernels), not "at most three of eight" 
    assert flipped <= FLIPPED_SEEDS, (sorted(flipped), sorted(FLIPPED_SEEDS))


@pytest.mark.parametrize("att", ["none", "mul"])
def test_gradient_norms_on_eight_unsearched_seeds(seeds_steps, att):
    from tests.c1w_cases import SEEDS

    res, gold = seeds_steps
    worst = (0.0, "")
    noise = 0.0
    for s in SEEDS:
        _, amax, norms = res[(att, s)]
        if (amax != gold[f"s{s}_argmax"].astype(np.int64)).any():
            continue  # (a tie pixel on the other region moves every gradient upstream of ConvPnPNet)
        for n, g in norms.items():
            ref = float(gold[f"s{s}_{att}_gradnorm/{n}"])
            noise = max(noise, float(gold[f"s{s}_{att}_gradnoise/{n}"]) if ref > 1e-4 else 0.0)
            if ref <= 1e-4:  # biases in front of a normalisation: the true gradient is zero
                assert g <= 1e-3, (s, n, g)
                continue
            e = abs(g - ref) / ref
            if e > worst[0]:
                worst = (e, f"seed {s} {n}")
            assert e <= 2e-2, (s, n, g, ref)
    print(f"[train seeds {att}] gradient norms of all parameters vs the reference: worst {worst[0]:.1e} ({worst[1]}); the reference's own "
          f"1-vs-8-thread gradient noise on these batches: up to {noise:.1e}")


@pytest.mark.parametrize("att,seed", [("none", 0), ("mul", 1), ("none", 1), ("mul", 0)])
def test_all_164_gradients_decision_forced_on_unsearched_seeds(golden_dir, att, seed):
    """the SHARP gradient test - every one of the 164 parameter gradients as a full tensor against the autograd of
    the reference-pinned oracle, with the HIP forward's 48 ReLU / LeakyReLU decisions (oracle.forced_relu_masks) and its region arg-max
    (force_argmax) forced in, bound 2e-4 - no longer lives on the one searched training batch of model_c1w.npz only
    (tests/test_gpu_c1w.py): here on unsearched seeds 0 and 1, both attention variants.  The losses of the forced oracle equal the
    reference's golden losses for the same seed (1e-5; pose-branch losses when no tie pixel flipped), which ties the forced oracle
    to the real reference on this very batch."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from rdpn6d_amd.train import TrainEngine
    from tests.c1w_cases import c1w_state_dict
    from tests.test_gpu_c1w import _hip_relu_masks

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "train_c1w_seeds.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention=att, device="cuda"))
    sdn = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sdn.items()}
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = synth.make_inputs(4, seed=seed)
    gt = synth.make_train_gt(4, inp)
    assert synth.sha256_of([inp[k] for k in sorted(inp)]) == str(gold[f"s{seed}_sha256_inputs"])
    eng = TrainEngine(model, 4, dev)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **gt}.items()}
    losses = {k: float(v.item()) for k, v in eng.forward_backward(batch).items()}
    torch.cuda.synchronize()
    amax = eng.argmax.cpu().numpy().reshape(4, 64, 64).astype(np.int64)
    flips = int((amax != gold[f"s{seed}_argmax"].astype(np.int64)).sum())
    grads = {n: p.grad.detach().cpu().double().clone() for n, p in model.named_parameters()}
    orc = model_oracle.GDRNOracle(32, att)
    orc.load_state_dict(sd, strict=True)
    orc.train()
    tc = {k: torch.from_numpy(v) for k, v in {**inp, **gt}.items()}
    with model_oracle.forced_relu_masks(orc, _hip_relu_masks(eng, orc)) as forced:
        oo = orc(tc["roi_img"], tc["roi_coord_2d"], tc["fps"], tc["roi_cam"], tc["roi_center"], tc["roi_wh"], tc["resize_ratio"],
                 train_pose=True, force_argmax=amax)
        olosses = model_oracle.gdrn_losses(oo, tc, tc["roi_extent"])
        sum(olosses.values()).backward()
    assert len(forced.used) == len(forced.masks) == 48
    for k, v in losses.items():
        ref = float(gold[f"s{seed}_{att}_{k}"])
        if k not in POSE or flips == 0:
            assert abs(v - ref) <= 1e-5 * max(1.0, abs(ref)), (k, v, ref)
        assert abs(float(olosses[k]) - v) <= 1e-5 * max(1.0, abs(v)), (k, float(olosses[k]), v)  # forced oracle vs HIP: same decisions
    rows = []
    for name, g in grads.items():
        ref = dict(orc.named_parameters())[name].grad.double()
        if ref.norm().item() < 1e-4:  # exact gradient zero up to round-off (a conv bias in front of a BatchNorm)
            assert g.norm().item() < 1e-4, name
            continue
        rows.append(((g - ref).norm().item() / ref.norm().item(), name))
    rows.sort(reverse=True)
    print(f"[train seeds {att} seed {seed}] HIP vs decision-forced oracle autograd, {len(rows)} tensors ({flips} tie px on the other region): "
          f"median {np.median([r[0] for r in rows]):.2e}, worst " + ", ".join(f"{n} {e:.2e}" for e, n in rows[:3]))
    assert len(rows) == 160
    for e, name in rows:
        assert e <= 2e-4, (name, e)
