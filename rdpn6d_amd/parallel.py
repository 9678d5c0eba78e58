"""Data parallelism of the hot path: one process per GPU, RCCL (``backend="nccl"``) over xGMI.

* Inference shards crops with NO collective on the measured path - the reference's ``InferenceSampler``
  contiguous split (core/utils/my_distributed_sampler.py:191-194) - plus an optional gather of the (B,12) poses.
* Training is pure data parallel: replicated weights, per-rank BatchNorm statistics, one
  gradient all-reduce per step.  The reference lets torch DDP bucket 164 tensors (25 MB buckets); here all
  gradients live in ONE flat HBM buffer laid out [pnp_net | rot_head_net | backbone] = the order in which the
  backward completes them, and three large all-reduces are issued as soon as each group is done, overlapping the
  rest of the backward (xGMI is point-to-point: few, large collectives).  BN buffers are not broadcast (each rank
  keeps its own statistics, a stated choice; the reference's DDP default broadcasts rank 0's).
"""
import torch
import torch.distributed as dist

GROUPS = ("pnp_net", "rot_head_net", "backbone")  # completion order of the backward


def flat_grad_storage(grads):
    """the 1-D fp32 tensor spanning the ONE storage all of `grads` live in (the flat gradient buffer of GradBuckets / Ranger), or
    None.  By storage rather than by ``_base``: a gradient autograd's AccumulateGrad adopted (torch DDP, gdrn._HipBackward) is a
    detached alias of the flat buffer's memory, not a view of the flat tensor."""
    grads = [g for g in grads if g is not None]
    if not grads or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads[1:]) or st.nbytes() % 4:
        return None
    if sum(g.numel() for g in grads) * 4 > st.nbytes():  # (aliases of each other rather than tiles of one buffer)
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, 0, (st.nbytes() // 4,))


def shard_range(n, rank, world):
    """contiguous shard [begin, end) of n items for this rank (InferenceSampler semantics)"""
    shard = (n - 1) // world + 1 if n > 0 else 0
    begin = shard * rank
    return min(begin, n), min(shard * (rank + 1), n)


class GradBuckets:
    """Flat gradient storage + bucketed all-reduce.  ``param.grad`` of every parameter becomes a view into one
    contiguous buffer; ``reduce(group)`` starts the (asynchronous) all-reduce of that group's slice,
    ``finish()`` waits for all of them and turns sums into means."""

    def __init__(self, model, groups=GROUPS, process_group=None, optimizer=None, always_reduce=False):
        """always_reduce: issue the all-reduces even in a one-rank group (a no-op on the values; the single-GPU RCCL test uses it to
        put real RCCL kernels between the backward and the optimizer).
        optimizer: the rdpn6d_amd Ranger that will step these parameters.  If it has already built its flat gradient
        buffer (a step was taken) that buffer is ADOPTED here - each group is a contiguous slice of it in any order - so both
        sides keep working on the same memory; an un-built Ranger adopts this buffer at its first step by itself."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())
        params = {g: [p for p in getattr(model, g).parameters() if p.requires_grad] for g in groups}
        self.params = params
        n = sum(p.numel() for ps in params.values() for p in ps)
        ref = next(p for ps in params.values() for p in ps)
        flat = getattr(optimizer, "_flat", None)
        flat = flat["g"] if flat else None
        self.slices = {}
        if flat is not None:
            if flat.numel() != n:
                raise ValueError("GradBuckets: the optimizer's flat gradient buffer does not cover exactly the model's trainable parameters")
            for g in groups:
                lo = min((p.grad.data_ptr() - flat.data_ptr()) // 4 for p in params[g])
                cnt = sum(p.numel() for p in params[g])
                inside = all(0 <= (p.grad.data_ptr() - flat.data_ptr()) // 4 - lo <= cnt - p.numel() for p in params[g])
                if not inside:
                    raise ValueError(f"GradBuckets: group {g!r} is not one contiguous slice of the optimizer's flat gradient buffer")
                self.slices[g] = (lo, lo + cnt)
            self.flat = flat
        else:
            self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
            o = 0
            for g in groups:
                b = o
                for p in params[g]:
                    p.grad = self.flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
                self.slices[g] = (b, o)
        self._home = {id(p): (p.grad.data_ptr() - self.flat.data_ptr()) // 4 for ps in params.values() for p in ps}
        self.handles = []
        # gloo has no AVG: reduce with SUM and scale in finish()
        self.avg_op = self.active and dist.get_backend(process_group) == "nccl"

    def _rehome(self, group):
        """a caller may have replaced param.grad since construction (model.zero_grad(set_to_none=True) followed by a
        backward that allocates fresh tensors): bring such gradients back into the flat buffer BEFORE it is reduced -
        reducing a buffer the gradients no longer live in would silently leave the ranks un-synchronised."""
        base = self.flat.data_ptr()
        for p in self.params[group]:
            o = self._home[id(p)]
            if p.grad is None:
                raise RuntimeError("GradBuckets.reduce: a parameter of group %r has no gradient (backward not run?)" % group)
            if p.grad.data_ptr() != base + 4 * o:
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view

    def zero_(self):
        self.flat.zero_()

    def reduce(self, group):
        self._rehome(group)
        if not self.active:
            return
        b, e = self.slices[group]
        op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
        self.handles.append(dist.all_reduce(self.flat[b:e], op=op, group=self.pg, async_op=True))

    def finish(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.active and not self.avg_op:
            self.flat.div_(self.world)


def gather_poses(rot, trans, process_group=None):
    """all ranks' (R|t) rows, rank-major: the (B,12)-float counterpart of the reference's pickle all_gather at the end
    of evaluation (gdrn_evaluator.py:440-441)"""
    pose = torch.cat([rot.reshape(rot.shape[0], 9), trans.reshape(trans.shape[0], 3)], 1).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return pose
    out = [torch.empty_like(pose) for _ in range(dist.get_world_size(process_group))]
    dist.all_gather(out, pose, group=process_group)
    return torch.cat(out, 0)


def reduce_loss_dict(loss_dict, process_group=None):
    """mean of the 9 scalar losses over ranks in ONE small all-reduce, no host sync
    (the reference: comm.reduce_dict + .item() per key, engine.py:299-300)"""
    names = sorted(loss_dict)
    v = torch.stack([loss_dict[k].detach().float() for k in names])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(v, group=process_group)
        v = v / dist.get_world_size(process_group)
    return dict(zip(names, v.unbind(0)))
