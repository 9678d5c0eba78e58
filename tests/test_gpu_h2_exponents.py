"""Per-tensor exponents of the h2 format (round 6): an h2 tensor holds a * 2^e, e = 4 by default (|a| < 4094); a
tensor that leaves that range has ITS exponent lowered - by the forward that overflowed it (one range-flag slot per launch names the
tensor) or beforehand by GDRN.calibrate_h2 - instead of the whole model dropping to the bf16x3 kernels at half the speed.  The
exponents live in the fp32 scale / shift vectors the host hands the kernels (powers of two: exact), the kernels are unchanged, so a plan
with every exponent at its default is the round-5 plan bit for bit (every other parity test of the suite runs on it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MAPS = ("mask", "coor_x", "coor_y", "coor_z", "region")


def _fwd(model, t):
    with torch.no_grad():
        o = model(t["roi_img"], roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"],
                  roi_whs=t["roi_wh"], roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}


def _weights(golden_dir):
    import os

    from oracle import model_oracle
    from rdpn6d_amd import synth

    orc = model_oracle.GDRNOracle(32, "none")
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in orc.state_dict().items()}, seed=1234)
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    sd.update({k: bn[k] for k in bn.files})
    return {k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd.items()}


def _heavy_tailed(sd, S=4096.0):
    """The same FUNCTION with activations thousands of times larger inside: a BatchNorm's (gamma, beta) times S and every consumer's
    weights divided by S (ReLU is positively homogeneous; S a power of two: exact in fp32) - what a checkpoint whose BN gammas grew
    looks like to the format.  Three places, one of each kind:
      * an ordinary activation: layer2.0's first conv -> BN -> ReLU output (consumer: layer2.0.conv2);
      * a whole RESIDUAL CHAIN: every tensor of layer3's identity path (layer3.0.downsample.1 and all bn2 of the stage; consumers: the
        stage's conv1 weights from block 1 on, layer4.0.conv1 and layer4.0.downsample.0) - the chain shares one exponent;
      * a head activation: rot_head_net.features.4 (consumer: features.6)."""
    sd = {k: v.clone() for k, v in sd.items()}

    def up(bn):
        sd[bn + ".weight"] *= S
        sd[bn + ".bias"] *= S

    def down(conv):
        sd[conv + ".weight"] /= S

    up("backbone.layer2.0.bn1"), down("backbone.layer2.0.conv2")
    nblk = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.layer3."))
    up("backbone.layer3.0.downsample.1")
    for i in range(nblk):
        up(f"backbone.layer3.{i}.bn2")
        if i:
            down(f"backbone.layer3.{i}.conv1")
    down("backbone.layer4.0.conv1"), down("backbone.layer4.0.downsample.0")
    up("rot_head_net.features.4"), down("rot_head_net.features.6")
    return sd


def _oracle64(sd, inp):
    from oracle import model_oracle

    orc = model_oracle.GDRNOracle(32, "none")
    orc.load_state_dict(sd, strict=True)
    orc = orc.double().eval()
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in inp.items()}
    t = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in t.items()}
    with torch.no_grad():
        return orc(t["roi_img"], t["roi_coord_2d"], t["fps"], t["roi_cam"], t["roi_center"], t["roi_wh"], t["resize_ratio"])


def _model(sd):
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    model, _ = build_model_optimizer(gdrn_base_cfg(mask_attention="none", device="cuda"))
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model


class _no_warnings:
    def __enter__(self):
        import warnings

        self._cm = warnings.catch_warnings()
        self._cm.__enter__()
        warnings.simplefilter("error", RuntimeWarning)

    def __exit__(self, *a):
        return self._cm.__exit__(*a)


def test_heavy_tailed_weights_stay_on_h2_inside_the_bare_tolerance(golden_dir):
    """activations ~1e3 .. 1e5 in three tensors (an activation, a residual chain, a head layer): the forward that meets them lowers
    exactly those tensors' exponents, re-runs the batch and hands out maps within 1e-4 of the float64 evaluation of the same weights -
    no warning, no switch to bf16x3; the next forward (and another batch size) runs straight through on the adapted plan."""
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    sd = _heavy_tailed(_weights(golden_dir))
    inp = synth.make_inputs(4, seed=5)
    want = _oracle64(sd, inp)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items()}
    model = _model(sd)
    with _no_warnings():
        o = _fwd(model, t)
    tab = dict(model.h2_exponents(dev))
    plan = model.plan(4, dev)
    assert model.cfg.TEST.FP16X2 is True and plan.fast == "h2" and not model.h2_range_exceeded(dev)
    assert set(tab) == {"layer2.0.conv1", "layer3", "rot_head.features.3"}, tab  # exactly the three tensors, nothing else
    assert all(e < 4 for e in tab.values()), tab
    err = {k: (o[k].cpu().double() - want[k]).abs().max().item() for k in MAPS}
    print("heavy-tailed weights on h2, exponents", tab, "max-abs vs float64:", {k: f"{v:.1e}" for k, v in err.items()})
    assert max(err.values()) < 1e-4, err
    er = max(((o["rot"][b].cpu().double() - want["rot"][b]).norm() / want["rot"][b].norm()).item() for b in range(4))
    et = max(((o["trans"][b].cpu().double() - want["trans"][b]).norm() / want["trans"][b].norm()).item() for b in range(4))
    assert er < 1e-4 and et < 1e-4, (er, et)
    # the adapted table serves the next forward without a re-plan, and a new batch size builds its plan from it
    with _no_warnings():
        o2 = _fwd(model, t)
        assert model.plan(4, dev) is plan
        o1 = _fwd(model, {k: v[:1].contiguous() for k, v in t.items()})
    assert all(torch.equal(o[k], o2[k]) for k in MAPS) and dict(model.h2_exponents(dev)) == tab
    assert max((o1[k][0] - o[k][0]).abs().max().item() for k in MAPS) < 1e-4 and not model.h2_range_exceeded(dev)


def test_calibrate_h2_sets_the_exponents_before_any_overflow(golden_dir):
    """GDRN.calibrate_h2 on a calibration batch: every h2 tensor's largest record is measured launch by launch and its exponent set
    one binade under the format's end; the forwards that follow never raise a flag, and the maps agree with float64 like above."""
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    sd = _heavy_tailed(_weights(golden_dir))
    inp = synth.make_inputs(4, seed=5)
    want = _oracle64(sd, inp)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items()}
    model = _model(sd)
    kw = dict(roi_classes=t["roi_cls"], roi_coord_2d=t["roi_coord_2d"], roi_cams=t["roi_cam"], roi_centers=t["roi_center"], roi_whs=t["roi_wh"],
              roi_extents=t["roi_extent"], resize_ratios=t["resize_ratio"], do_loss=False, fps=t["fps"])
    tab = model.calibrate_h2(t["roi_img"], **kw)
    assert {"layer2.0.conv1", "layer3", "rot_head.features.3"} <= set(tab) and all(e <= 4 for e in tab.values())
    plan = model.plan(4, dev)
    flag = model.h2_range_flag(dev)
    assert int(flag.abs().sum()) == 0
    with _no_warnings():
        o = _fwd(model, t)
    assert model.plan(4, dev) is plan and int(flag.abs().sum()) == 0 and dict(model.h2_exponents(dev)) == tab
    # one binade of head-room: the largest record of every calibrated tensor sits in [2^14, 2^15) or its exponent is the default
    err = max((o[k].cpu().double() - want[k]).abs().max().item() for k in MAPS)
    print("calibrated exponents:", {k: v for k, v in sorted(tab.items()) if v != 4}, f"maps max-abs vs float64 {err:.1e}")
    assert err < 1e-4


def test_lowered_exponents_cost_round_off_only(golden_dir):
    """the exponent is bookkeeping, not arithmetic: with EVERY variable two binades down on the ordinary fixture (activations O(1))
    the plan is still h2, nothing is flagged, and the maps stay inside the bare 1e-4 of the float64 evaluation.  They are NOT
    bit-identical: an h2 record's lo term is an fp16 subnormal below |a| * 2^e = 2^-3, so small activations keep a few bits less at a
    smaller exponent (absolute error 2^-25 / 2^e) - fp32-ulp-sized differences, which this fixture amplifies ~1e3-fold like any other
    round-off (measured: 2.9e-5 between e = 4 and e = 2).  That is why the default stays at 4 and tensors are lowered one at a time."""
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    sd = _weights(golden_dir)
    inp = synth.make_inputs(4, seed=5)
    want = _oracle64(sd, inp)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items()}
    model = _model(sd)
    base = _fwd(model, t)
    plan = model.plan(4, dev)
    names = sorted({v for v in plan._slots if v is not None})
    assert {"stem", "layer2", "layer3", "layer4", "spatial_net.emb", "spatial_net.l1", "spatial_net.l2", "spatial_net.l3", "rot_head.convT"} <= set(names)
    model.h2_exponents(dev).update({v: 2 for v in names})
    low = _fwd(model, t)
    assert model.plan(4, dev) is not plan and model.plan(4, dev).fast == "h2" and not model.h2_range_exceeded(dev)
    d = max((low[k] - base[k]).abs().max().item() for k in MAPS)
    e_base = max((base[k].cpu().double() - want[k]).abs().max().item() for k in MAPS)
    e_low = max((low[k].cpu().double() - want[k]).abs().max().item() for k in MAPS)
    print(f"all {len(names)} exponents 4 -> 2: maps move by {d:.1e}; max-abs vs float64 {e_base:.1e} (default) -> {e_low:.1e}")
    assert e_base < 1e-4 and e_low < 1e-4 and d < 1e-4


def test_deferred_range_check_adapts_the_exponents_at_the_next_forward(golden_dir):
    """cfg.TEST.H2_RANGE_CHECK = "deferred" (pipelined serving, no host wait per forward): the forward that overflowed returns clamped
    values; the NEXT forward finds the flag slots, lowers those tensors' exponents (a warning says that an earlier forward's outputs
    were computed with clamped values) and stays on h2; after at most a few such forwards the outputs are the adapted plan's."""
    from rdpn6d_amd import synth

    dev = torch.device("cuda:0")
    sd = _heavy_tailed(_weights(golden_dir))
    inp = synth.make_inputs(4, seed=5)
    want = _oracle64(sd, inp)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in inp.items()}
    model = _model(sd)
    model.cfg.TEST.H2_RANGE_CHECK = "deferred"
    with _no_warnings():
        _fwd(model, t)  # clamped, nobody waited
    assert model.h2_range_exceeded(dev, wait=True) and dict(model.h2_exponents(dev)) == {}
    warned = 0
    for _ in range(6):
        import warnings

        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            o = _fwd(model, t)
        warned += sum("EARLIER forward" in str(x.message) for x in w)
        torch.cuda.synchronize()
        if not model.h2_range_exceeded(dev, wait=True):
            break
    tab = dict(model.h2_exponents(dev))
    assert warned >= 1 and model.cfg.TEST.FP16X2 is True and model.plan(4, dev).fast == "h2"
    assert set(tab) == {"layer2.0.conv1", "layer3", "rot_head.features.3"} and all(e < 4 for e in tab.values()), tab
    assert max((o[k].cpu().double() - want[k]).abs().max().item() for k in MAPS) < 1e-4
