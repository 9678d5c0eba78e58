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


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Linear(5, 7)
        self.rot_head_net = nn.Conv2d(2, 3, 3)
        self.pnp_net = nn.Linear(4, 2, bias=False)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = _Toy()
    gb = GradBuckets(m)
    assert gb.flat.numel() == sum(p.numel() for p in m.parameters())
    order = []
    for g in ("pnp_net", "rot_head_net", "backbone"):  # the backward completes groups in this order
        for p in getattr(m, g).parameters():
            p.grad.fill_(float(rank + 1))
            p.grad.view(-1)[0] = 10.0 * (rank + 1)
        gb.reduce(g)
        order.append(g)
    gb.finish()
    ok = all(torch.allclose(p.grad.view(-1)[1:], torch.full((p.numel() - 1,), 1.5)) and abs(p.grad.view(-1)[0].item() - 15.0) < 1e-6
             for p in m.parameters())
    ok = ok and all(p.grad.data_ptr() >= gb.flat.data_ptr() for p in m.parameters())  # views into the flat buffer
    poses = gather_poses(torch.full((3, 3, 3), float(rank)), torch.full((3, 3), float(rank)))
    ok = ok and poses.shape == (6, 12) and poses[:3].eq(0).all().item() and poses[3:].eq(1).all().item()
    red = reduce_loss_dict({"loss_a": torch.tensor(float(rank)), "loss_b": torch.tensor(2.0 + rank)})
    ok = ok and abs(red["loss_a"].item() - 0.5) < 1e-6 and abs(red["loss_b"].item() - 2.5) < 1e-6
    b, e = shard_range(9, rank, world)
    ok = ok and (b, e) == ((0, 5) if rank == 0 else (5, 9))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


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
        from rdpn6d_amd.parallel import GROUPS

        for g in GROUPS:
            self.log.append("stage:" + g)
            for p in getattr(self.model, g).parameters():
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
        for name, p in m.named_parameters():
            p.register_post_accumulate_grad_hook(lambda p, name=name: eng.log.append("hook:" + name.split(".")[0]))
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
            for g, nxt in (("pnp_net", "rot_head_net"), ("rot_head_net", "backbone")):
                hooks = [i for i, e in enumerate(kinds) if e == "hook:" + g]
                if not hooks or not (pos["stage:" + g] < min(hooks) and max(hooks) < pos["stage:" + nxt]):
                    ok, msg = False, f"it {it}: hooks of {g} did not fire between its stage and the next: {kinds}"
            if sum(e.startswith("hook:") for e in kinds) != 5:
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
