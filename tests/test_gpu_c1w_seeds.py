"""SEED-INDEPENDENT parity at the bare north-star tolerances: the well-conditioned fixture on EIGHT CONSECUTIVE
input seeds (0..7, no search) = 32 crops captured from the REAL reference by tools/oracle/gen_model_golden_seeds.py, on the B = 64
plan bench.py times (two shuffled copies of the 32 crops), for all three fp32-accurate plans:

    h2   - two fp16 planes per operand, three exact partial products (the default, the headline's ``dtype: "f32"``)
    x3   - three bf16 planes, six partial products
    none - the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, undisputed fp32)

Asserted per plan: dense maps <= 1e-4 max-abs on every crop; ZERO region arg-max flips outside the RECORDED tie set (reference's own
top-2 logit gap < 2e-4, or a pixel the reference flips against itself between 1 and 8 threads / fp32 and float64 - the rule is in
the fixture, tests/c1w_cases.tie_set); pose <= 1e-4 on all 64 slots for the default plan h2, both attention variants; for the two
fall-back plans <= 3e-4 with the crops over 1e-4 being the RECORDED ones (OVER_BARE), each with its pose branch alone inside 1e-4.  A crop whose arg-max differs from the reference's at a tie pixel is compared with
the reference-pinned oracle restarted from the reference's golden maps with OUR region decision at those pixels (one flipped pixel
moves the reference's own pose by 3e-4 .. 3e-3: seed 2 crop 0, seed 3 crop 3 of the generator's log).
And ACROSS plans: h2's worst map / pose error <= 1.25 x the fp32-MFMA plan's on this fixture - the end-to-end proof that the
22-bit operands of h2 are not what the parity margin is spent on."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MAPS = ("mask", "coor_x", "coor_y", "coor_z", "region")
# crops (seed, index in the seed's batch) whose end-to-end rotation error is over the bare 1e-4 on the two fall-back plans, as measured
# on the round-5 code (deterministic kernels: the same on every box); the default plan has none
OVER_BARE = {("x3", "none"): set(), ("x3", "mul"): {(0, 2), (6, 3)}, ("none", "none"): {(7, 0)}, ("none", "mul"): {(5, 0), (6, 3)}}
FAST = {"h2": dict(BF16X3=True, FP16X2=True), "x3": dict(BF16X3=True, FP16X2=False), "none": dict(BF16X3=False, FP16X2=True)}


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def seeds_run(golden_dir):
    """every (plan, attention) forward of the 64-slot batch, evaluated once"""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer
    from tests.c1w_cases import SEEDS, c1w_state_dict, tie_set

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "model_c1w_seeds.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    assert tuple(int(s) for s in gold["seeds"]) == SEEDS
    inps = []
    for s in SEEDS:
        inp = synth.make_inputs(4, seed=s)
        assert synth.sha256_of([inp[k] for k in sorted(inp)]) == str(gold[f"s{s}_sha256_inputs"]), s
        inps.append(inp)
    inp32 = {k: np.concatenate([i[k] for i in inps], 0) for k in inps[0]}
    crop = [(s, c) for s in SEEDS for c in range(4)]  # crop id -> (seed, index in the seed's batch)
    order = np.concatenate([np.arange(32), np.random.default_rng(5).permutation(32)])  # 64 slots: every crop twice
    t = {k: torch.from_numpy(np.ascontiguousarray(v[order])).to(dev) for k, v in inp32.items()}
    ref = {k: np.concatenate([gold[f"s{s}_{k}"] for s in SEEDS], 0) for k in MAPS + ("argmax",)}
    ref["tie"] = np.concatenate([tie_set(gold, s) for s in SEEDS], 0)
    for k in MAPS:  # the reference's own float64 maps (stored as the difference from its fp32 maps, x 2^14, float16)
        ref["exact_" + k] = ref[k].astype(np.float64) + np.concatenate([gold[f"s{s}_fp64diff_{k}"] for s in SEEDS], 0).astype(np.float64) / 16384.0
    for att in ("none", "mul"):
        for q in ("rot", "trans", "fp64err_rot", "fp64err_trans"):
            ref[f"{att}_{q}"] = np.concatenate([gold[f"s{s}_{att}_{q}"] for s in SEEDS], 0)
    out, sd = {}, None
    for att in ("none", "mul"):
        model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention=att, device="cuda"))
        if sd is None:
            sdn = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
            assert synth.sha256_of([sdn[k] for k in sorted(sdn) if not k.endswith("num_batches_tracked")]) == str(gold["sha256_weights"])
            sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sdn.items()}
        model.load_state_dict(sd, strict=True)
        model.eval()
        for kv in filter(None, os.environ.get("RDPN6D_SEEDS_TEST_CFG", "").split(",")):  # A/B runs: "FUSE_HEAD_OUT=0,PNP_H2=0"
            model.cfg.TEST[kv.split("=")[0].strip()] = bool(int(kv.split("=")[1]))
        for fast, sw in FAST.items():
            model.cfg.TEST.BF16X3, model.cfg.TEST.FP16X2 = sw["BF16X3"], sw["FP16X2"]
            with torch.no_grad():
                o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"],
                          roi_centers=t["roi_center"], roi_whs=t["roi_wh"], roi_extents=t["roi_extent"],
                          resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
            torch.cuda.synchronize()
            plan = model.plan(64, dev)
            assert plan.fast == (None if fast == "none" else fast), (fast, plan.fast)
            assert not model.h2_range_exceeded(dev) and model.cfg.TEST.FP16X2 == sw["FP16X2"]
            res = {k: o[k].cpu().numpy().astype(np.float64) for k in MAPS + ("rot", "trans")}
            res["argmax"] = plan.argmax.cpu().numpy().reshape(64, 64, 64)
            out[(fast, att)] = res
            model.invalidate_plans()  # (2.3 GB of activation buffers per plan)
            torch.cuda.empty_cache()
    # the reference-pinned oracle for the pose branch of crops that flip a tie pixel (restarted from the reference's golden maps)
    orcs = {}
    for att in ("none", "mul"):
        orc = model_oracle.GDRNOracle(32, att)
        orc.load_state_dict(sd, strict=True)
        orcs[att] = orc.eval()
    return out, ref, order, crop, inp32, orcs


def _pose_given_our_decisions(orc, ref, inp32, cid, amax, our_maps=None):
    """the oracle's pose branch on the REFERENCE's maps of crop `cid` (or, with `our_maps`, on the maps the plan itself produced for
    that slot), region decisions `amax` (64, 64) forced"""
    sl = slice(cid, cid + 1)
    ti = {k: torch.from_numpy(np.ascontiguousarray(v[sl])) for k, v in inp32.items()}
    maps = tuple(torch.from_numpy(np.ascontiguousarray(ref[k][sl] if our_maps is None else our_maps[k][None].astype(np.float32)))
                 for k in MAPS)
    with torch.no_grad():
        o = orc(ti["roi_img"], ti["roi_coord_2d"], ti["fps"], ti["roi_cam"], ti["roi_center"], ti["roi_wh"], ti["resize_ratio"],
                dense_maps=maps, force_argmax=amax[None])
    return o["rot"][0].numpy().astype(np.float64), o["trans"][0].numpy().astype(np.float64)


def _score(seeds_run, fast, att):
    out, ref, order, crop, inp32, orcs = seeds_run
    res = out[(fast, att)]
    worst = {k: float(np.abs(res[k] - ref[k][order].astype(np.float64)).max()) for k in MAPS}
    per_crop_map = np.max([np.abs(res[k] - ref[k][order].astype(np.float64)).reshape(64, -1).max(1) for k in MAPS], axis=0)
    # ... and against the EXACT answer (the reference evaluated in float64): worst element and rms over all map elements
    exact_worst = max(float(np.abs(res[k] - ref["exact_" + k][order]).max()) for k in MAPS)
    exact_rms = float(np.sqrt(np.mean(np.concatenate([((res[k] - ref["exact_" + k][order]) ** 2).reshape(-1) for k in MAPS]))))
    ref_exact_worst = max(float(np.abs(ref[k].astype(np.float64) - ref["exact_" + k]).max()) for k in MAPS)
    ref_exact_rms = float(np.sqrt(np.mean(np.concatenate([((ref[k].astype(np.float64) - ref["exact_" + k]) ** 2).reshape(-1) for k in MAPS]))))
    diff = res["argmax"] != ref["argmax"][order]
    outside = int((diff & ~ref["tie"][order]).sum())
    inside = int((diff & ref["tie"][order]).sum())
    er, et, bare, forced, tols, forced_rows, given_maps = [], [], 0, 0, [], [], {}
    for slot in range(64):
        cid = int(order[slot])
        R, T = ref[f"{att}_rot"][cid].astype(np.float64), ref[f"{att}_trans"][cid].astype(np.float64)
        if diff[slot].any():  # a tie pixel took the other region: the reference's pose GIVEN that decision
            plain = _rel(res["rot"][slot], R)
            R, T = _pose_given_our_decisions(orcs[att], ref, inp32, cid, res["argmax"][slot])
            forced += 1
            forced_rows.append((slot, crop[cid], int(diff[slot].sum()), plain, _rel(res["rot"][slot], R)))
        e_r, e_t = _rel(res["rot"][slot], R), _rel(res["trans"][slot], T)
        tol = 1e-4 if att == "none" else min(2e-4, max(1e-4, 1.5 * float(ref[f"{att}_fp64err_rot"][cid])))
        er.append(e_r), et.append(e_t), tols.append(tol)
        if e_r > 1e-4:  # an ill-conditioned crop: the pose branch alone, i.e. against the reference-pinned oracle fed OUR maps
            Rg, _ = _pose_given_our_decisions(orcs[att], ref, inp32, cid, res["argmax"][slot], {k: res[k][slot] for k in MAPS})
            given_maps[slot] = (crop[cid], float(per_crop_map[slot]), e_r, _rel(res["rot"][slot], Rg))
        bare += e_r <= 1e-4 and e_t <= 1e-4
    return dict(worst=worst, per_crop_map=per_crop_map, outside=outside, inside=inside, er=np.asarray(er), et=np.asarray(et), bare=bare,
                forced=forced, tols=np.asarray(tols), given_maps=given_maps, ntie=int(ref["tie"][order].sum()), forced_rows=forced_rows, exact_worst=exact_worst,
                exact_rms=exact_rms, ref_exact_worst=ref_exact_worst, ref_exact_rms=ref_exact_rms)


@pytest.mark.parametrize("att", ["none", "mul"])
@pytest.mark.parametrize("fast", ["h2", "x3", "none"])
def test_bare_tolerances_on_eight_unsearched_seeds(seeds_run, fast, att):
    s = _score(seeds_run, fast, att)
    print(f"[seeds {fast} {att}] maps max-abs over 64 slots: " + " ".join(f"{k} {v:.2e}" for k, v in s["worst"].items())
          + f" | arg-max flips outside the tie set {s['outside']}, inside {s['inside']} (tie set: {s['ntie']} of {64 * 4096} pixels; "
          f"{s['forced']} slots compared given our decision) | pose worst R {s['er'].max():.2e} t {s['et'].max():.2e}; "
          f"slots within the BARE 1e-4: {s['bare']} / 64")
    for slot, (seed, c), npx, plain, given in s["forced_rows"]:
        print(f"      slot {slot} (seed {seed} crop {c}): {npx} tie pixel(s) took the other region; rotation vs the reference's pose {plain:.2e}, "
              f"vs the reference's pose GIVEN our decision {given:.2e}")
    for slot, ((seed, c), dmap, e2e, given) in s["given_maps"].items():
        print(f"      slot {slot} (seed {seed} crop {c}) is over the bare 1e-4 end to end: maps {dmap:.2e} from the reference's, rotation {e2e:.2e}; "
              f"pose branch alone (oracle fed our maps) {given:.2e}")
    for k, v in s["worst"].items():
        assert v <= 1e-4, (k, v)
    assert s["outside"] == 0
    # wherever a pose is over the bare tolerance end to end, the pose branch itself is inside it: what is left is the reference's own
    # sensitivity to a <= 1e-4 change of its maps on that crop
    assert all(g[3] <= 1e-4 for g in s["given_maps"].values()), s["given_maps"]
    over = {crop for crop, _, _, _ in s["given_maps"].values()}  # crops (seed, index) whose rotation is over the bare 1e-4 end to end
    if fast == "h2":
        # the default plan, the one bench.py times, BOTH attention variants: every slot inside the bare 1e-4 (measured: worst R 5.6e-5 none /
        # 5.3e-5 mul, t 2.2e-5) - round 4 accepted up to 2e-4 under "mul", where a regression to 1.98e-4 once hid
        assert s["bare"] == 64 and s["er"].max() <= 1e-4 and s["et"].max() <= 1e-4, (s["bare"], s["er"].max(), s["et"].max())
    else:
        # the two fall-back plans: maps inside 1e-4 like h2's (asserted above), pose inside 3e-4, and the crops over the bare 1e-4 are
        # the RECORDED ones - not a count.  Each of them has its pose branch alone inside 1e-4 (asserted above):
        # what is left is the reference's own sensitivity to a <= 6e-5 change of its maps on that crop.
        assert s["er"].max() <= 3e-4 and s["et"].max() <= 1e-4, (s["er"].max(), s["et"].max())
        assert over <= OVER_BARE[(fast, att)], (fast, att, sorted(over), sorted(OVER_BARE[(fast, att)]))
        assert s["bare"] == 64 - 2 * len(over)  # (every crop sits in two slots)


def test_h2_is_as_accurate_as_the_fp32_mfma_plan_end_to_end(seeds_run):
    """h2 holds each operand as two fp16 terms (22 bits); the claim that the headline is fp32-ACCURATE is checked where it counts:
    on the whole network, next to the undisputed fp32-MFMA plan on the same 64 slots - against the real reference's fp32 maps / poses
    AND against the EXACT answer (the reference's own float64 evaluation, stored in the fixture), because the reference's fp32 maps
    are themselves ~5e-5 from exact and a distance to them mixes both errors.
    Asserted: pose (what the path delivers) - h2 no worse than 1.25 x the fp32-MFMA plan, worst slot and mean, both attention
    variants; maps vs exact - h2's rms error within 1.25 x and its worst element within 1.5 x the fp32-MFMA plan's (measured on the
    round-4 box: see the printed table; h2 stores 22-bit activations, the fp32 pipe 24-bit ones, and the worst single element over
    9.7 M is the noisiest statistic of the lot) and NEITHER plan further from exact than 2 x the reference's own fp32 evaluation."""
    rows = {f: _score(seeds_run, f, "none") for f in FAST}
    rows_mul = {f: _score(seeds_run, f, "mul") for f in FAST}
    r0 = rows["h2"]
    print(f"[seeds] the REFERENCE's own fp32 maps vs its float64 evaluation: worst {r0['ref_exact_worst']:.2e} rms {r0['ref_exact_rms']:.2e}")
    for f in FAST:
        print(f"[seeds {f}] maps vs reference fp32: worst {max(rows[f]['worst'].values()):.2e} (mean of per-crop worst {rows[f]['per_crop_map'].mean():.2e}) | maps vs EXACT: "
              f"worst {rows[f]['exact_worst']:.2e} rms {rows[f]['exact_rms']:.2e} | pose none: worst R {rows[f]['er'].max():.2e} mean {rows[f]['er'].mean():.2e} | pose mul: "
              f"worst R {rows_mul[f]['er'].max():.2e} mean {rows_mul[f]['er'].mean():.2e}")
    h, n = rows["h2"], rows["none"]
    assert h["er"].max() <= 1.25 * n["er"].max() and h["er"].mean() <= 1.25 * n["er"].mean()
    assert rows_mul["h2"]["er"].max() <= 1.25 * rows_mul["none"]["er"].max() and rows_mul["h2"]["er"].mean() <= 1.25 * rows_mul["none"]["er"].mean()
    assert h["exact_rms"] <= 1.25 * n["exact_rms"] and h["exact_worst"] <= 1.5 * n["exact_worst"], (h["exact_rms"], n["exact_rms"], h["exact_worst"], n["exact_worst"])
    for f in FAST:
        assert rows[f]["exact_worst"] <= 2.0 * r0["ref_exact_worst"] and rows[f]["exact_rms"] <= 2.0 * r0["ref_exact_rms"], f
