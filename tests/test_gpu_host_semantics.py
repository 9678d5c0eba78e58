"""GPU tests of the host-side contracts around the kernels (the reference's training / evaluation loop semantics):

  * evaluation after training steps sees the NEW weights (engine.py: TEST.EVAL_PERIOD, sanity eval before training);
  * Ranger <-> GradBuckets share ONE flat gradient buffer whatever the group order, and survive the reference loop's
    ``optimizer.zero_grad(set_to_none=True)`` (engine.py:304);
  * ``backward`` honours the gradient autograd hands to the losses (GradScaler scale, 1/accum, weighted / partial sums);
  * two data-parallel ranks through the REAL TrainEngine.backward(on_group_done=buckets.reduce) + Ranger: buckets fire in
    completion order, every gradient is the rank mean, weights stay identical across ranks (engine.py:292-313).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(att="mul", seed=1234):
    from rdpn6d_amd import synth
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.gdrn import build_model_optimizer

    model, opt = build_model_optimizer(gdrn_base_cfg(mask_attention=att, device="cuda"))
    sd = synth.make_trained_like_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return model, opt


def _batch(B, seed, dev):
    from rdpn6d_amd import synth

    inp = synth.make_inputs(B, seed=seed)
    return {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(B, inp)}.items()}


def _eval(model, b):
    model.eval()
    with torch.no_grad():
        o = model(b["roi_img"], roi_classes=b["roi_cls"], roi_coord_2d=b["roi_coord_2d"], roi_cams=b["roi_cam"], roi_centers=b["roi_center"],
                  roi_whs=b["roi_wh"], roi_extents=b["roi_extent"], resize_ratios=b["resize_ratio"], do_loss=False, fps=b["fps"])
    torch.cuda.synchronize()
    return o


def _train_losses(model, b):
    model.train()
    od, ld = model(b["roi_img"], gt_xyz=b["roi_xyz"], gt_mask_trunc=b["roi_mask_trunc"], gt_mask_visib=b["roi_mask_visib"],
                   gt_mask_obj=b["roi_mask_obj"], gt_region=b["roi_region"], gt_ego_rot=b["ego_rot"], gt_points=b["roi_points"],
                   sym_infos=None, gt_trans=b["trans"], gt_trans_ratio=b["roi_trans_ratio"], roi_classes=b["roi_cls"],
                   roi_coord_2d=b["roi_coord_2d"], roi_cams=b["roi_cam"], roi_centers=b["roi_center"], roi_whs=b["roi_wh"],
                   roi_extents=b["roi_extent"], resize_ratios=b["resize_ratio"], do_loss=True, fps=b["fps"])
    assert od == {}
    return ld


def test_eval_after_training_steps_uses_the_new_weights():
    """eval -> two optimizer steps -> eval: the second evaluation must come from the stepped weights and the updated
    BatchNorm statistics (the fused Ranger step and the BN update write through raw pointers, so nothing but the weights
    epoch tells the cached InferencePlan that its packed copies are stale), and must equal a fresh model loaded from the
    trained state_dict bit for bit."""
    dev = torch.device("cuda:0")
    model, opt = _model()
    b = _batch(2, 3, dev)
    o1 = _eval(model, b)
    maps1, rot1 = o1["region"].clone(), o1["rot"].clone()
    for g in opt.param_groups:
        g["lr"] = 1e-2
    for _ in range(2):
        ld = _train_losses(model, b)
        opt.zero_grad(set_to_none=True)
        sum(ld.values()).backward()
        opt.step()
    o2 = _eval(model, b)
    assert torch.equal(o1["region"], maps1), "outputs of an earlier forward must not be overwritten by a later one"
    assert (o2["region"] - maps1).abs().max().item() > 1e-3 and (o2["rot"] - rot1).abs().max().item() > 1e-4
    fresh, _ = _model()
    fresh.load_state_dict(model.state_dict(), strict=True)
    o3 = _eval(fresh, b)
    for k in ("mask", "coor_x", "region", "rot", "trans"):
        assert torch.equal(o2[k], o3[k]), k
    # a plain torch in-place edit is seen too (tensor version counters)
    with torch.no_grad():
        model.rot_head_net.features[21].bias.add_(1.0)
    o4 = _eval(model, b)
    assert abs((o4["region"] - o2["region"]).mean().item() - 1.0) < 1e-4


@pytest.mark.parametrize("amp", [False, "bf16"])
def test_backward_stages_yield_final_gradients_in_completion_order(amp):
    """the backbone's gradient bucket is split per ResNet stage.  TrainEngine.backward_stages() must yield
    parallel.STAGES in order - pnp_net -> rot_head_net -> backbone.layer4 -> backbone.layer3 -> backbone.rest - and at each yield the
    stage's parameter gradients must be FINAL: a snapshot taken there (what an all-reduce issued there would read) equals the
    gradient after the whole backward, bit for bit.  Under AMP the stage's grouped weight gradient (rdpn6d_wgrad_bf16_group) is one
    of the launches that must have run by then."""
    from rdpn6d_amd.parallel import STAGES, stage_params

    dev = torch.device("cuda:0")
    model, _ = _model()
    if amp:
        model.cfg.SOLVER.AMP.ENABLED, model.cfg.SOLVER.AMP.DTYPE = True, amp
    eng = model.train_engine(2, dev)
    b = _batch(2, 4, dev)
    for p in model.parameters():
        p.grad = torch.full_like(p, float("nan"))  # the backward WRITES every gradient: nothing of this may survive
    eng.forward_losses(b)
    eng.seed_backward({n: 1.0 for n in eng.LOSS_NAMES})
    seen, snaps = [], {}
    for st in eng.backward_stages():
        torch.cuda.synchronize()
        seen.append(st)
        snaps[st] = [p.grad.clone() for p in stage_params(model, st)]
    torch.cuda.synchronize()
    assert seen == list(STAGES)
    for st in STAGES:
        ps = stage_params(model, st)
        assert len(ps) == len(snaps[st]) > 0
        for p, g in zip(ps, snaps[st]):
            assert torch.isfinite(g).all(), st
            assert torch.equal(p.grad, g), f"a gradient of stage {st} was still written after the stage was handed on"
    assert sum(len(v) for v in snaps.values()) == len(list(model.parameters()))


def test_ranger_and_gradbuckets_share_one_buffer_in_any_order():
    """the factory's optimizer orders its groups backbone | rot_head | pnp_net, GradBuckets lays the gradients out
    pnp_net | rot_head_net | backbone; zero_grad(set_to_none=True) runs before the first step as in engine.py:304.  The
    gradients must stay views of the bucket buffer, the fused step must read exactly that buffer, and the result must be
    bit-identical to a run without buckets."""
    from rdpn6d_amd.parallel import GradBuckets

    dev = torch.device("cuda:0")
    b = _batch(2, 4, dev)
    runs = {}
    for with_buckets in (False, True):
        model, opt = _model()
        buckets = GradBuckets(model, optimizer=opt) if with_buckets else None
        for it in range(2):
            ld = _train_losses(model, b)
            opt.zero_grad(set_to_none=True)
            sum(ld.values()).backward()
            if buckets is not None:
                for g in ("pnp_net", "rot_head_net", "backbone"):
                    buckets.reduce(g)
                buckets.finish()
                lo, hi = buckets.flat.data_ptr(), buckets.flat.data_ptr() + 4 * buckets.flat.numel()
                assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
            opt.step()
            if buckets is not None:
                assert opt._flat["g"].data_ptr() == buckets.flat.data_ptr(), "Ranger must adopt the bucket buffer, not copy out of it"
        torch.cuda.synchronize()
        runs[with_buckets] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k in runs[False]:
        assert torch.equal(runs[False][k], runs[True][k]), k
    # the other construction order: optimizer already built (one step taken), buckets attached afterwards adopt ITS buffer
    model, opt = _model()
    ld = _train_losses(model, b)
    opt.zero_grad(set_to_none=True)
    sum(ld.values()).backward()
    opt.step()
    buckets = GradBuckets(model, optimizer=opt)
    assert buckets.flat.data_ptr() == opt._flat["g"].data_ptr()
    from rdpn6d_amd.parallel import STAGES, stage_params

    assert tuple(buckets.slices) == STAGES
    for g, (lo, hi) in buckets.slices.items():  # every stage is one contiguous slice of the buffer the optimizer laid out
        assert hi - lo == sum(p.numel() for p in stage_params(model, g))
    # a caller that drops the gradients (nn.Module.zero_grad defaults to set_to_none=True) is re-homed before the reduce
    ld = _train_losses(model, b)
    model.zero_grad(set_to_none=True)
    sum(ld.values()).backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    for g in ("pnp_net", "rot_head_net", "backbone"):
        buckets.reduce(g)
    buckets.finish()
    lo, hi = buckets.flat.data_ptr(), buckets.flat.data_ptr() + 4 * buckets.flat.numel()
    for n, p in model.named_parameters():
        assert lo <= p.grad.data_ptr() < hi and torch.equal(p.grad, want[n]), n


def test_backward_honours_the_upstream_gradient():
    """GradScaler-style scaling (engine.py:302-309), gradient accumulation and weighted / partial sums of the loss dict."""
    dev = torch.device("cuda:0")
    model, _ = _model()
    b = _batch(2, 5, dev)

    def grads(total_of):
        model.zero_grad(set_to_none=True)
        total_of(_train_losses(model, b)).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    g1 = grads(lambda ld: sum(ld.values()))
    gs = grads(lambda ld: sum(ld.values()) * 65536.0)     # the scale GradScaler starts with
    ga = grads(lambda ld: sum(ld.values()) / 4.0)         # accumulation over 4 micro-batches
    for n in g1:
        assert torch.equal(gs[n], g1[n] * 65536.0) and torch.equal(ga[n], g1[n] * 0.25), n  # powers of two: exact
    xyz = ("loss_coor_x", "loss_coor_y", "loss_coor_z")
    gx = grads(lambda ld: sum(ld[k] for k in xyz))                                        # a partial sum
    gr = grads(lambda ld: sum(v for k, v in ld.items() if k not in xyz))
    gw = grads(lambda ld: 3.0 * sum(ld[k] for k in xyz) + sum(v for k, v in ld.items() if k not in xyz))
    worst = 0.0
    for n in g1:
        den = g1[n].double().norm().item()
        if den < 1e-4:
            continue
        worst = max(worst, ((gx[n] + gr[n]).double() - g1[n].double()).norm().item() / den,
                    ((3.0 * gx[n] + gr[n]).double() - gw[n].double()).norm().item() / max(gw[n].double().norm().item(), 1e-30))
    print(f"linearity of the backward in the loss weights: worst relative deviation {worst:.2e}")
    assert worst < 1e-4
    assert gx["pnp_net.fc1.weight"].abs().max().item() == 0.0  # the dense xyz losses do not depend on ConvPnPNet's weights
    with pytest.raises(NotImplementedError):
        grads(lambda ld: ld["loss_coor_x"] + 2.0 * ld["loss_coor_y"])
    with pytest.raises(FloatingPointError):
        grads(lambda ld: sum(ld.values()) * float("inf"))


def test_loss_scale_and_micro_batch_accumulation_do_not_compound():
    """(a) seed_backward is absolute - seeding twice does not compound the factor, and a second backward through the same
    node (retain_graph) raises; (b) forward_backward with a loss scale un-scales only what THIS backward produced, so with
    `accumulate_grad` the gradients of earlier micro-batches are not divided again; (c) without accumulate_grad a backward WRITES
    param.grad (the documented contract: the reference loop zeroes before every backward, engine.py:304-308)."""
    dev = torch.device("cuda:0")
    model, opt = _model("none")
    b1, b2 = _batch(2, 11, dev), _batch(2, 12, dev)
    eng = model.train_engine(2, dev)

    def grads_of(batch, scale=1.0):
        eng.loss_scale = scale
        eng.forward_backward(batch)
        torch.cuda.synchronize()
        return torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()]).clone()

    opt.zero_grad(set_to_none=True)
    g1 = grads_of(b1)
    g2 = grads_of(b2)                     # (c) overwrites: equals b2's own gradients, not g1 + g2
    model2, _ = _model("none")
    eng2 = model2.train_engine(2, dev)
    eng2.forward_backward(b1)             # same BatchNorm-statistics history as `model` had when it saw b2
    eng2.forward_backward(b2)
    torch.cuda.synchronize()
    assert torch.equal(g2, torch.cat([p.grad.detach().reshape(-1) for p in model2.parameters()]))
    # (a) absolute seeds: scale 4096 via seed_backward called twice + forward_backward's own call
    eng.loss_scale = 1.0
    eng.forward_losses(b2)
    eng.seed_backward({n: 4096.0 for n in eng.LOSS_NAMES})
    eng.seed_backward({n: 4096.0 for n in eng.LOSS_NAMES})
    eng.backward(unscale=4096.0)
    torch.cuda.synchronize()
    g2s = torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()]).clone()
    assert torch.equal(g2s, g2)  # same weights, same batch (training-mode BatchNorm uses batch statistics): 4096 drops out exactly
    ld = _train_losses(model, b2)
    tot = sum(ld.values())
    opt.zero_grad(set_to_none=True)
    tot.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="retain_graph is not supported"):  # the HIP backward works in place: a second pass is refused, not silently wrong
        (tot * 8.0).backward()
    # (b) two micro-batches with a loss scale, accumulated: g(b1) + g(b2) of the SAME weights / statistics sequence without a scale
    def two_micro(scale, m):
        e = m.train_engine(2, dev)
        e.accumulate_grad, e.loss_scale = True, scale
        have = [p.grad for p in m.parameters() if p.grad is not None]
        if have:
            torch._foreach_zero_(have)  # (in place: the engine's launches hold the gradients' addresses)
        e.forward_backward(b1)
        e.forward_backward(b2)
        torch.cuda.synchronize()
        e.accumulate_grad = False
        return torch.cat([p.grad.detach().reshape(-1) for p in m.parameters()]).clone()

    ma, _ = _model("none")
    mb, _ = _model("none")
    acc1, acc_s = two_micro(1.0, ma), two_micro(1024.0, mb)
    assert torch.equal(acc1, acc_s), "power-of-two loss scale must drop out exactly, also for the first micro-batch"
    mc, _ = _model("none")
    ec = mc.train_engine(2, dev)
    ec.forward_backward(b1)
    torch.cuda.synchronize()
    first = torch.cat([p.grad.detach().reshape(-1) for p in mc.parameters()]).clone()
    ec.forward_backward(b2)
    torch.cuda.synchronize()
    second = torch.cat([p.grad.detach().reshape(-1) for p in mc.parameters()])
    assert torch.equal(acc1, first + second)


# ----------------------------------------------------------------------------- two data-parallel ranks on one GPU
def _dp_worker(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist

    from rdpn6d_amd.parallel import GradBuckets, reduce_loss_dict

    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        # RCCL cannot put two ranks on one device: two ranks rendezvous over gloo, the RCCL run has one rank
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        model, opt = _model()  # factory optimizer: Ranger, groups backbone | rot_head | pnp_net
        b = _batch(2, 100 + rank, dev)  # every rank its own crops
        eng = model.train_engine(2, dev)
        # (1) stand-alone gradients of this rank (no reduction), gathered on the host
        eng.forward_backward(b)
        torch.cuda.synchronize()
        mine = torch.cat([p.grad.detach().reshape(-1).cpu() for p in model.parameters()])
        if backend == "nccl":
            mean = mine
        else:
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            mean = sum(both) / world
        model, opt = _model()  # fresh BatchNorm statistics / weights for the data-parallel run
        eng = model.train_engine(2, dev)
        buckets = GradBuckets(model, optimizer=opt, always_reduce=backend == "nccl")
        fired = []

        def on_done(g):
            fired.append(g)
            buckets.reduce(g)

        ok, msg = True, ""
        for it in range(3):
            eng.refresh_weights()
            losses = eng.forward_losses(b)
            opt.zero_grad(set_to_none=True)
            eng.backward(on_group_done=on_done)
            buckets.finish()
            if it == 0:
                torch.cuda.synchronize()
                got = torch.cat([p.grad.detach().reshape(-1).cpu() for p in model.parameters()])
                err = ((got - mean).abs().max() / mean.abs().max()).item()
                if fired != ["pnp_net", "rot_head_net", "backbone.layer4", "backbone.layer3", "backbone.rest"] or buckets.last_issue_order != fired:
                    ok, msg = False, f"bucket order {fired}"
                if err > 1e-6:
                    ok, msg = False, f"reduced gradient differs from the rank mean: {err:.2e}"
            red = reduce_loss_dict(losses)
            opt.step()
            if opt._flat["g"].data_ptr() != buckets.flat.data_ptr():
                ok, msg = False, "Ranger steps a different buffer than the one that was reduced"
        torch.cuda.synchronize()
        w = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])
        if backend == "nccl":
            # the same three steps without any process group: the RCCL all-reduce of one rank (its own stream, async handles waited on
            # before Ranger) must leave every bit of the trajectory alone - a missing stream dependency shows up here
            model, opt = _model()
            eng = model.train_engine(2, dev)
            for it in range(3):
                eng.refresh_weights()
                eng.forward_losses(b)
                opt.zero_grad(set_to_none=True)
                eng.backward()
                opt.step()
            torch.cuda.synchronize()
            ws = [w, torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])]
            if dist.get_backend() != "nccl" or not buckets.active:
                ok, msg = False, "the reduction did not go through RCCL"
        else:
            ws = [torch.empty_like(w) for _ in range(world)]
            dist.all_gather(ws, w)
        if not torch.equal(ws[0], ws[1]):
            ok, msg = False, f"weights diverged across ranks: {(ws[0] - ws[1]).abs().max().item():.3e}"
        if len(red) != 9 or not all(torch.isfinite(v) for v in red.values()):
            ok, msg = False, "loss reduction"
        q.put((rank, ok, msg))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, False, traceback.format_exc()[-1500:] + repr(e)))


def test_two_rank_data_parallel_training_through_the_real_engine():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res


def test_rccl_backed_gradient_reduction_through_the_real_engine():
    """The same loop over RCCL (backend "nccl"): one rank - RCCL refuses two ranks on one device and this box has one GPU - so the
    all-reduces really run as RCCL kernels on RCCL's stream against the flat gradient buffer, between the backward kernels and the
    fused Ranger step: bucket order, reduced == stand-alone gradients, and a three-step trajectory bit-identical to the run without
    a process group (what a missing stream dependency or a reduction of the wrong buffer would break)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    p = ctx.Process(target=_dp_worker, args=(0, 1, port, q, "nccl"))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[:2] == (0, True), res


def _ddp_worker(rank, world, port, q, backend="gloo", bucket_view=False):
    """the factory's model wrapped in torch DistributedDataParallel, driven by the reference loop LITERALLY (engine.py:292-309:
    forward through the wrapper, sum of the loss dict, optimizer.zero_grad(set_to_none=True), losses.backward(), optimizer.step())"""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        b = _batch(2, 100 + rank, dev)  # every rank its own crops
        # (1) stand-alone gradients of this rank (no wrapper), gathered on the host -> the rank mean DDP must produce
        model, opt = _model()
        ld = _train_losses(model, b)
        opt.zero_grad(set_to_none=True)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        mine = torch.cat([p.grad.detach().reshape(-1).cpu() for p in model.parameters()])
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        mean = sum(both) / world
        # (2) the wrapped run
        model, opt = _model()
        fired = []
        from rdpn6d_amd.parallel import STAGES, stage_params

        for g in STAGES:
            stage_params(model, g)[0].register_post_accumulate_grad_hook(lambda p, g=g: fired.append(g))
        ddp = DDP(model, device_ids=[0], gradient_as_bucket_view=bucket_view)
        ok, msg = True, ""
        for it in range(3):
            ddp.train()
            od, loss_dict = ddp(b["roi_img"], gt_xyz=b["roi_xyz"], gt_mask_trunc=b["roi_mask_trunc"], gt_mask_visib=b["roi_mask_visib"],
                                gt_mask_obj=b["roi_mask_obj"], gt_region=b["roi_region"], gt_ego_rot=b["ego_rot"], gt_points=b["roi_points"],
                                sym_infos=None, gt_trans=b["trans"], gt_trans_ratio=b["roi_trans_ratio"], roi_classes=b["roi_cls"],
                                roi_coord_2d=b["roi_coord_2d"], roi_cams=b["roi_cam"], roi_centers=b["roi_center"], roi_whs=b["roi_wh"],
                                roi_extents=b["roi_extent"], resize_ratios=b["resize_ratio"], do_loss=True, fps=b["fps"])
            losses = sum(loss_dict.values())
            opt.zero_grad(set_to_none=True)
            losses.backward()
            if it == 0:
                torch.cuda.synchronize()
                got = torch.cat([p.grad.detach().reshape(-1).cpu() for p in model.parameters()])
                err = ((got - mean).abs().max() / mean.abs().max()).item()
                if err > 1e-6:
                    ok, msg = False, f"DDP-reduced gradient differs from the rank mean: {err:.2e}"
                if fired != list(STAGES):
                    ok, msg = False, f"AccumulateGrad hooks fired in the order {fired}"
                if not bucket_view:  # the gradients never left Ranger's-to-be flat layout: still one tensor per parameter, written once
                    if any(p.grad is None for p in model.parameters()):
                        ok, msg = False, "a parameter has no gradient"
            opt.step()
        torch.cuda.synchronize()
        w = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])
        ws = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        if not torch.equal(ws[0], ws[1]):
            ok, msg = False, f"weights diverged across ranks: {(ws[0] - ws[1]).abs().max().item():.3e}"
        if (w - torch.cat([p.detach().reshape(-1).cpu() for p in _model()[0].parameters()])).abs().max().item() == 0.0:
            ok, msg = False, "the weights did not move"
        q.put((rank, ok, msg))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, False, traceback.format_exc()[-2500:] + repr(e)))


@pytest.mark.parametrize("bucket_view", [False, True])
def test_torch_ddp_wrapper_around_the_factory_model_through_the_reference_loop(bucket_view):
    """VERDICT r3 item 2 / SURVEY 8b "the module is wrapped by DDP/_LiteModule" (main_gdrn.py:113, engine.py:308): two ranks (gloo;
    they share the box's one GPU, which RCCL refuses) wrap the factory's model in torch DDP and run the reference loop literally for
    three steps on different crops.  The chained autograd nodes of gdrn._attach_hip_backward hand every parameter gradient to
    autograd, so DDP's AccumulateGrad hooks fire (group by group, in backward-completion order), every gradient equals the mean of
    the two ranks' stand-alone gradients to 1e-6, the second and third iteration do not trip DDP's "expected to have finished
    reduction" check, and the weights stay bit-identical across ranks.  Both DDP gradient layouts (copy / bucket views)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + os.getpid() % 300 + (7 if bucket_view else 0)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, "gloo", bucket_view)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res


def test_train_vis_scalars_on_device_match_the_reference_and_never_sync_per_step(golden_dir):
    """VERDICT r3 item 7 / SURVEY section 3: the `vis/*` scalars the reference's train forward pushes to EventStorage
    (GDRN.py:306-368: compute_mean_re_te + sixteen `.item()` reads of crop 0) - kept, computed by ONE small kernel per step into a
    device table, copied to the host once per cfg.TRAIN.VIS_PERIOD steps.  Checked: (1) every delivered row equals the
    reference-pinned numpy restatement evaluated on the engine's own train-mode pose to 1e-5; (2) the first row equals the values
    the REAL reference pushed on this batch (golden: vis_scalars_golden.npz) within the train-mode pose tolerance; (3) rows arrive
    in blocks of N through `model.vis_sink`, the first block only after step N; (4) off by default."""
    from oracle import model_oracle
    from rdpn6d_amd import synth
    from tests.c1w_cases import c1w_state_dict

    dev = torch.device("cuda:0")
    gold = np.load(os.path.join(golden_dir, "vis_scalars_golden.npz"))
    bn = np.load(os.path.join(golden_dir, "bn_stats_c1w.npz"))
    inp = synth.make_inputs(4, seed=int(gold["train_input_seed"]))
    b = {k: torch.from_numpy(v).to(dev) for k, v in {**inp, **synth.make_train_gt(4, inp)}.items()}
    for att in ("none", "mul"):
        model, opt = _model(att)
        sd = c1w_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, bn)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        got, want = [], []
        model.vis_sink = got.append
        _train_losses(model, b)
        assert got == [] and "_vis" not in model.__dict__, "VIS_SCALARS must be off by default"
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        model.cfg.TRAIN.VIS_SCALARS, model.cfg.TRAIN.VIS_PERIOD = True, 2
        for it in range(5):
            ld = _train_losses(model, b)
            eng = model.train_engine(4, dev)
            want.append(model_oracle.train_vis_scalars(eng.trans.cpu(), eng.rot.cpu(), eng.rt[:, 6:9].cpu(), b["trans"].cpu(), b["ego_rot"].cpu(),
                                                       b["roi_trans_ratio"].cpu()))
            assert 2 * max(0, (it + 1) // 2 - 1) <= len(got) <= 2 * (it // 2) and len(got) % 2 == 0, (it, len(got))  # whole blocks, one block behind at most
            opt.zero_grad(set_to_none=True)
            sum(ld.values()).backward()
            opt.step()
        model.flush_vis_scalars()
        assert len(got) == 5 and len(model.vis_history) == 5
        for row, ref in zip(got, want):
            assert tuple(row) == model.VIS_NAMES == model_oracle.VIS_NAMES
            for k in model.VIS_NAMES:
                assert abs(row[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (att, k, row[k], ref[k])
        for k in model.VIS_NAMES:  # the REAL reference's values for this batch (its train-mode pose differs from ours by ~1e-5)
            ref = float(gold[f"{att}_{k}"])
            assert abs(got[0][k] - ref) <= 2e-4 * max(1.0, abs(ref)), (att, k, got[0][k], ref)
        print(f"[vis {att}] error_R {got[0]['vis/error_R']:.4f} deg (reference {float(gold[att + '_vis/error_R']):.4f}), error_t "
              f"{got[0]['vis/error_t']:.4f} cm (reference {float(gold[att + '_vis/error_t']):.4f})")


def _overflowing(b):
    """the batch with its image 10^5 times too bright: the stem's activations leave the +-4094 of the h2 format"""
    o = dict(b)
    o["roi_img"] = b["roi_img"] * 1.0e5
    return o


def _want_bf16x3(b):
    ref, _ = _model("none")
    ref.cfg.TEST.FP16X2 = False
    return _eval(ref, b)


@pytest.mark.parametrize("graph", [False, True])
def test_h2_range_overflow_is_caught_in_the_forward_that_overflowed(graph):
    """an activation beyond the fp16 range of the h2 kernels must not reach the caller as a clamped result.  Default
    (cfg.TEST.H2_RANGE_CHECK = "sync"): the forward reads the device flag for ITSELF, warns, switches the model to the bf16x3
    kernels and re-runs the batch - also when the forward is a replayed hipGraph captured on an input that did not overflow,
    also when the overflow happens on a plan (batch size) that is used only once, and the switch survives a plan rebuild."""
    dev = torch.device("cuda:0")
    b, b3 = _batch(2, 3, dev), _batch(3, 5, dev)
    hot = _overflowing(b)
    want = _want_bf16x3(hot)
    model, _ = _model("none")
    model.cfg.TEST.HIP_GRAPH = graph
    with warnings_none():
        for _ in range(3):  # (graph: eager, capture, replay) on data that stays in range
            o_ok = _eval(model, b)
    assert model.cfg.TEST.FP16X2 is True and not model.h2_range_exceeded(dev) and model.plan(2, dev).fast == "h2"
    if graph:
        assert any(not isinstance(g, str) for g in model.plan(2, dev)._graphs.values()), "the graph was not captured"
        keep = b["roi_img"].clone()
        b["roi_img"].copy_(hot["roi_img"])  # same buffers -> the captured graph replays on data that now overflows
        hot = b
    with pytest.warns(RuntimeWarning, match="exceeded"):
        o = _eval(model, hot)
    assert model.cfg.TEST.FP16X2 is False
    for k in ("mask", "coor_x", "region", "rot", "trans"):
        assert torch.isfinite(o[k]).all() and torch.equal(o[k], want[k]), k
    with warnings_none():
        o3 = _eval(model, b3)  # another batch size, new plan: stays on bf16x3, no second warning
        assert model.plan(3, dev).fast == "x3" and torch.isfinite(o3["rot"]).all()
    assert torch.isfinite(o_ok["rot"]).all()


def test_h2_range_overflow_deferred_mode_is_reported_by_the_next_forward_of_any_plan():
    """cfg.TEST.H2_RANGE_CHECK = "deferred" (pipelined serving: no host wait per forward): the flag is model-level, so the NEXT
    forward sees it whatever its batch size, and h2_range_exceeded() answers at any sync point."""
    dev = torch.device("cuda:0")
    b, b3 = _batch(2, 3, dev), _batch(3, 5, dev)
    model, _ = _model("none")
    model.cfg.TEST.H2_RANGE_CHECK = "deferred"
    with warnings_none():
        o = _eval(model, _overflowing(b))  # clamped values, no wait, no warning yet
    assert model.cfg.TEST.FP16X2 is True and torch.isfinite(o["rot"]).all()
    assert model.h2_range_exceeded(dev, wait=True)
    with pytest.warns(RuntimeWarning, match="earlier forward"):
        o3 = _eval(model, b3)
    assert model.cfg.TEST.FP16X2 is False and model.plan(3, dev).fast == "x3"
    want3 = _want_bf16x3(b3)
    assert torch.equal(o3["rot"], want3["rot"]) and torch.equal(o3["region"], want3["region"])


class warnings_none:
    """context: any RuntimeWarning inside is an error"""

    def __enter__(self):
        import warnings

        self._cm = warnings.catch_warnings()
        self._cm.__enter__()
        warnings.simplefilter("error", RuntimeWarning)

    def __exit__(self, *a):
        return self._cm.__exit__(*a)
