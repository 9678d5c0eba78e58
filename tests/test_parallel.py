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
