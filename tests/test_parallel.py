"""world_size-2 gloo tests (CPU) of the data-parallel host logic: contiguous inference shards, the flat bucketed
gradient all-reduce, pose gather and loss reduction."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from rdpn6d_amd.parallel import GradBuckets, gather_poses, reduce_loss_dict, shard_range


def test_shard_range_matches_inference_sampler():
    for n in (0, 1, 7, 64, 65, 512):
        for world in (1, 2, 3, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            idx = [i for b, e in got for i in range(b, e)]
            assert idx == list(range(n))
            if n:
                shard = (n - 1) // world + 1
                assert all(e - b <= shard for b, e in got)


class _ToyTrunk(nn.Module):
    """the sub-module names (and registration order) of gdrn's backbone holder: spatial_net first, then the stem, then the stages"""

    def __init__(self):
        super().__init__()
        self.spatial_net = nn.Linear(3, 2)
        self.conv1 = nn.Linear(5, 7)
        self.bn1 = nn.BatchNorm1d(7)
        self.layer1 = nn.Linear(2, 2, bias=False)
        self.layer2 = nn.Linear(4, 3)
        self.layer3 = nn.Linear(3, 6)
        self.layer4 = nn.Linear(6, 9)


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = _ToyTrunk()
        self.rot_head_net = nn.Conv2d(2, 3, 3)
        self.pnp_net = nn.Linear(4, 2, bias=False)


def _fill(m, stage, rank, salt=0.0):
    from rdpn6d_amd.parallel import stage_params

    for i, p in enumerate(stage_params(m, stage)):
        p.grad.copy_(torch.arange(p.numel(), dtype=torch.float32).view_as(p) * (0.37 + rank) + (i + 1) * 1.1 ** rank + salt)


def _worker(rank, world, port, q):
    from rdpn6d_amd.parallel import GROUPS, STAGES

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = _Toy()
    gb = GradBuckets(m)  # one bucket per stage
    assert gb.flat.numel() == sum(p.numel() for p in m.parameters())
    assert gb.groups == STAGES == ("pnp_net", "rot_head_net", "backbone.layer4", "backbone.layer3", "backbone.rest")
    for g in STAGES:  # the backward completes the stages in this order
        for p in stage_params_of(m, g):
            p.grad.fill_(float(rank + 1))
            p.grad.view(-1)[0] = 10.0 * (rank + 1)
        gb.reduce(g)
    ok = gb._issued == list(STAGES)
    gb.finish()
    ok = ok and gb.last_issue_order == list(STAGES)
    ok = ok and all(torch.allclose(p.grad.view(-1)[1:], torch.full((p.numel() - 1,), 1.5)) and abs(p.grad.view(-1)[0].item() - 15.0) < 1e-6
                    for p in m.parameters())
    ok = ok and all(p.grad.data_ptr() >= gb.flat.data_ptr() for p in m.parameters())  # views into the flat buffer
    # every stage is one contiguous slice, laid out in completion order
    ends = [gb.slices[g] for g in STAGES]
    ok = ok and ends[0][0] == 0 and all(ends[i][1] == ends[i + 1][0] for i in range(4)) and ends[-1][1] == gb.flat.numel()
    # --- five buckets == three buckets, bit for bit (round 6: the backbone's bucket split per ResNet stage)
    means = {}
    for form, groups in (("stages", STAGES), ("coarse", GROUPS)):
        m2 = _Toy()
        g2 = GradBuckets(m2, groups=groups)
        issued_at = []
        for st in STAGES:
            _fill(m2, st, rank)
            g2.reduce(st)
            issued_at.append(list(g2._issued))
        if form == "coarse":  # the coarse backbone bucket waits for its LAST stage
            ok = ok and issued_at == [["pnp_net"], ["pnp_net", "rot_head_net"], ["pnp_net", "rot_head_net"], ["pnp_net", "rot_head_net"],
                                      list(GROUPS)]
        g2.finish()
        means[form] = {n: p.grad.clone() for n, p in m2.named_parameters()}
    ok = ok and all(torch.equal(means["stages"][n], means["coarse"][n]) for n in means["stages"])
    # ... and both are the rank mean of what the two ranks wrote
    m3 = _Toy()
    g3 = GradBuckets(m3)
    want = {}
    for r in range(world):
        for st in STAGES:
            _fill(m3, st, r)
        for n, p in m3.named_parameters():
            want[n] = want.get(n, 0) + p.grad.clone()
    ok = ok and all(torch.equal(means["stages"][n], want[n] / world) for n in want)
    # --- bf16 transport (cfg.SOLVER.ALLREDUCE_DTYPE): the mean of the bf16-rounded gradients, inside one bf16 rounding of the fp32 mean
    m4 = _Toy()
    g4 = GradBuckets(m4, comm_dtype=torch.bfloat16)
    for st in STAGES:
        _fill(m4, st, rank)
        g4.reduce(st)
    g4.finish()
    for n, p in m4.named_parameters():
        ref = want[n] / world
        ok = ok and p.grad.dtype == torch.float32 and bool(((p.grad - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-6).all())
    # --- a bucket nobody reduced is an error, not a silent divergence of the ranks
    m5 = _Toy()
    g5 = GradBuckets(m5)
    g5.reduce("pnp_net")
    try:
        g5.finish()
        ok = False
    except RuntimeError as e:
        ok = ok and "never reduced" in str(e)
    g5.reduce("pnp_net"), g5.reduce("rot_head_net"), g5.reduce("backbone")  # (a coarse name = all of its stages; same collectives on both ranks)
    g5.finish()
    poses = gather_poses(torch.full((3, 3, 3), float(rank)), torch.full((3, 3), float(rank)))
    ok = ok and poses.shape == (6, 12) and poses[:3].eq(0).all().item() and poses[3:].eq(1).all().item()
    red = reduce_loss_dict({"loss_a": torch.tensor(float(rank)), "loss_b": torch.tensor(2.0 + rank)})
    ok = ok and abs(red["loss_a"].item() - 0.5) < 1e-6 and abs(red["loss_b"].item() - 2.5) < 1e-6
    b, e = shard_range(9, rank, world)
    ok = ok and (b, e) == ((0, 5) if rank == 0 else (5, 9))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def stage_params_of(m, g):
    from rdpn6d_amd.parallel import stage_params

    return stage_params(m, g)


def test_bucketed_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_single_process_is_a_no_op():
    m = _Toy()
    gb = GradBuckets(m)
    for p in m.parameters():
        p.grad.fill_(2.0)
    gb.reduce("backbone")
    gb.finish()
    assert all((p.grad == 2).all() for p in m.parameters())
    assert gb.last_issue_order == ["backbone.layer4", "backbone.layer3", "backbone.rest"]


def test_stage_layout_and_order():
    """parallel.STAGES = the order the backward completes parameter gradients in: pnp -> head -> layer4 -> layer3 ->
    rest; each stage is one contiguous slice of the buffer GradBuckets builds AND of a buffer in module.parameters() order (what an
    optimizer that built first hands over); spatial_net (done early, tiny) rides with the tail."""
    from rdpn6d_amd.parallel import GROUPS, STAGES, stage_params, stages_of

    assert STAGES == ("pnp_net", "rot_head_net", "backbone.layer4", "backbone.layer3", "backbone.rest")
    assert stages_of("backbone") == STAGES[2:] and stages_of("pnp_net") == ("pnp_net",)
    with pytest.raises(KeyError):
        stages_of("neck")
    m = _Toy()
    rest = stage_params(m, "backbone.rest")
    names = {id(p): n for n, p in m.named_parameters()}
    assert [names[id(p)].split(".")[1] for p in rest] == ["spatial_net", "spatial_net", "conv1", "conv1", "bn1", "bn1", "layer1", "layer2", "layer2"]
    assert sum(p.numel() for g in STAGES for p in stage_params(m, g)) == sum(p.numel() for p in m.parameters())
    assert [id(p) for p in stage_params(m, "backbone")] == [id(p) for st in STAGES[2:] for p in stage_params(m, st)]
    # parameters() order: [spatial_net conv1 bn1 layer1 layer2 | layer3 | layer4] - every stage contiguous there too
    order = [id(p) for p in m.backbone.parameters()]
    for st in STAGES[2:]:
        idx = sorted(order.index(id(p)) for p in stage_params(m, st))
        assert idx == list(range(idx[0], idx[0] + len(idx))), st
    with pytest.raises(ValueError):
        GradBuckets(m, groups=("pnp_net", "backbone"))  # rot_head_net missing
    m.backbone.layer3.weight.requires_grad_(False)  # frozen parameters are left out of the buckets
    gb = GradBuckets(m, groups=GROUPS)
    assert gb.flat.numel() == sum(p.numel() for p in m.parameters() if p.requires_grad)
    pose = gather_poses(torch.zeros(2, 3, 3), torch.ones(2, 3))
    assert pose.shape == (2, 12)


# ---------------------------------------------------------------------------------------------------------------------
# torch DDP around the model (main_gdrn.py:113 `self.setup(model, optimizer)`, engine.py:308 `self.backward(losses)`): the chained
# autograd nodes of gdrn._attach_hip_backward must make every parameter's AccumulateGrad - DDP's hook point - fire, group by group.
# CPU part: the node chain with a stand-in engine (the real TrainEngine needs the GPU: tests/test_gpu_host_semantics.py).
class _FakeEngine:
    LOSS_NAMES = ("loss_coor_x", "loss_coor_y", "loss_coor_z", "loss_mask", "loss_region", "loss_region_my", "loss_PM_R", "loss_centroid", "loss_z")

    def __init__(self, model, rank):
        self.model, self.rank, self.dev, self.log, self.seed = model, rank, torch.device("cpu"), [], None

    def seed_backward(self, w):
        self.seed = w

    def backward_stages(self):
        from rdpn6d_amd.parallel import STAGES, stage_params

        for g in STAGES:
            self.log.append("stage:" + g)
            for p in stage_params(self.model, g):
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.copy_(torch.full_like(p, (self.rank + 1.0) * self.seed["loss_mask"]))  # WRITES, like the kernels
            yield g


class _ToyLoss(_Toy):
    def forward(self, x):
        from rdpn6d_amd.gdrn import _attach_hip_backward

        losses = {n: torch.tensor(float(i)) for i, n in enumerate(self.eng.LOSS_NAMES)}
        return {}, _attach_hip_backward(self, self.eng, losses)


def _ddp_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(0)
        m = _ToyLoss()
        m.eng = eng = _FakeEngine(m, rank)
        from rdpn6d_amd.parallel import STAGES, stage_params

        stage_of = {id(p): st for st in STAGES for p in stage_params(m, st)}
        for name, p in m.named_parameters():
            p.register_post_accumulate_grad_hook(lambda p: eng.log.append("hook:" + stage_of[id(p)]))
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        ok, msg = True, ""
        for it in range(3):  # the reference loop, literally (engine.py:292-309); a second iteration is where DDP raises if a hook was missed
            eng.log.clear()
            _, loss_dict = ddp(torch.zeros(1))
            losses = sum(loss_dict.values())
            opt.zero_grad(set_to_none=True)
            (losses * 2.0).backward()
            want = 2.0 * (1.0 + 2.0) / 2.0  # upstream gradient 2 x the mean of (rank + 1) over the two ranks
            for p in m.parameters():
                if p.grad is None or not torch.allclose(p.grad, torch.full_like(p, want)):
                    ok, msg = False, f"it {it}: gradient is not the rank mean: {None if p.grad is None else p.grad.reshape(-1)[:3]}"
            kinds = [e for e in eng.log]
            # every group's hooks fire right after ITS stage and before the next stage starts
            pos = {e: i for i, e in enumerate(kinds) if e.startswith("stage:")}
            for g, nxt in zip(STAGES[:-1], STAGES[1:]):
                hooks = [i for i, e in enumerate(kinds) if e == "hook:" + g]
                if not hooks or not (pos["stage:" + g] < min(hooks) and max(hooks) < pos["stage:" + nxt]):
                    ok, msg = False, f"it {it}: hooks of {g} did not fire between its stage and the next: {kinds}"
            if sum(e.startswith("hook:") for e in kinds) != len(list(m.parameters())):
                ok, msg = False, f"it {it}: {kinds}"
            opt.step()
        w = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
        ws = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        if not torch.equal(ws[0], ws[1]):
            ok, msg = False, "weights diverged"
        # no_grad forward: plain tensors, no graph
        with torch.no_grad():
            _, ld = m(torch.zeros(1))
        if any(v.requires_grad for v in ld.values()):
            ok, msg = False, "no_grad forward built a graph"
        q.put((rank, ok, msg))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, False, traceback.format_exc()[-2000:] + repr(e)))


def test_ddp_hooks_fire_through_the_chained_backward_nodes_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res


def test_gradients_handed_to_autograd_stay_in_the_flat_buffer():
    """the stage takes the flat-buffer view out of param.grad and returns it; AccumulateGrad adopts it WITHOUT a copy, so Ranger's /
    GradBuckets' flat buffer still holds the gradient - and is still recognised (by storage) as the one flat buffer"""
    from rdpn6d_amd.parallel import flat_grad_storage

    m = _ToyLoss()
    m.eng = _FakeEngine(m, 0)
    gb = GradBuckets(m)
    before = {n: p.grad.data_ptr() for n, p in m.named_parameters()}
    _, ld = m(torch.zeros(1))
    sum(ld.values()).backward()
    for n, p in m.named_parameters():
        assert p.grad.data_ptr() == before[n], n
        assert (p.grad == 1.0).all()
    assert (gb.flat == 1.0).all()
    flat = flat_grad_storage([p.grad for p in m.parameters()])
    assert flat is not None and flat.data_ptr() == gb.flat.data_ptr() and flat.numel() == gb.flat.numel()
    assert flat_grad_storage([torch.zeros(3), torch.zeros(4)]) is None
    with pytest.raises(RuntimeError):  # frozen everything -> nothing to differentiate
        for p in m.parameters():
            p.requires_grad_(False)
        m(torch.zeros(1))


def test_gradbuckets_from_cfg_reads_the_solver_keys():
    """cfg.SOLVER.ALLREDUCE_DTYPE (f32 | bf16) and cfg.SOLVER.GRAD_BUCKETS (stages | coarse): the two switches of the gradient exchange"""
    from rdpn6d_amd.config import gdrn_base_cfg
    from rdpn6d_amd.parallel import GROUPS, STAGES

    cfg = gdrn_base_cfg(device="cpu")
    gb = GradBuckets.from_cfg(_Toy(), cfg)
    assert gb.groups == STAGES and gb.comm_dtype is None
    cfg.SOLVER.ALLREDUCE_DTYPE, cfg.SOLVER.GRAD_BUCKETS = "bf16", "coarse"
    gb = GradBuckets.from_cfg(_Toy(), cfg)
    assert gb.groups == GROUPS and gb.comm_dtype is torch.bfloat16
    cfg.SOLVER.ALLREDUCE_DTYPE = "fp8"
    with pytest.raises(ValueError, match="ALLREDUCE_DTYPE"):
        GradBuckets.from_cfg(_Toy(), cfg)
