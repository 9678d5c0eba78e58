"""Data parallelism of the hot path: one process per GPU, RCCL (``backend="nccl"``) over xGMI.

* Inference shards crops with NO collective on the measured path - the reference's ``InferenceSampler``
  contiguous split (core/utils/my_distributed_sampler.py:191-194) - plus an optional gather of the (B,12) poses.
* Training is pure data parallel: replicated weights, per-rank BatchNorm statistics, one
  gradient all-reduce per step.  The reference lets torch DDP bucket 164 tensors (25 MB buckets); here all
  gradients live in ONE flat HBM buffer laid out in the order in which the backward completes them -
  STAGES = [pnp_net | rot_head_net | backbone.layer4 | backbone.layer3 | backbone.rest] (ResNet-34: 38 / 24 / 52 / 27 / 6 MB of
  fp32) - and one large all-reduce per stage is issued as soon as that stage's gradients are done, overlapping the rest of the
  backward (xGMI is point-to-point: few, large collectives).  Round 6 split the backbone: as one bucket its 86 MB (59 % of the
  bytes) were handed over only when the whole backward had finished - fully exposed; now layer4's 52 MB go out while layer3 ..
  stem are still being differentiated and only the last 6 MB are issued at the end.  BN buffers are not broadcast (each rank
  keeps its own statistics, a stated choice; the reference's DDP default broadcasts rank 0's).
"""
import torch
import torch.distributed as dist

GROUPS = ("pnp_net", "rot_head_net", "backbone")  # the three sub-modules, in the order the backward completes them (coarse buckets)
# What TrainEngine.backward_stages() yields, in order: a stage = a set of parameters whose gradients are complete at that point of the
# backward.  The ResNet stages' grouped weight gradients are issued with the stage's first block (the last one differentiated), so
# layer4's parameters are final before layer3's input gradients start.  "backbone.rest" = layer2, layer1, the stem and the point-wise
# fusion branch (spatial_net: done first, 0.2 M parameters - kept with the tail so that every stage is ONE contiguous slice of
# backbone.parameters() order as well: [spatial_net conv1 bn1 layer1 layer2 | layer3 | layer4]).
STAGES = ("pnp_net", "rot_head_net", "backbone.layer4", "backbone.layer3", "backbone.rest")


def stages_of(name):
    """the stages that make up bucket `name` (a stage itself, or a coarse group such as "backbone"), in completion order"""
    st = tuple(s for s in STAGES if s == name or s.startswith(name + "."))
    if not st:
        raise KeyError(f"unknown gradient stage / bucket {name!r}; stages are {STAGES}")
    return st


def stage_params(model, name, trainable_only=True):
    """parameters of a stage or bucket, in the order they lie in a flat gradient buffer (stage by stage, module order inside)"""
    out = []
    for s in stages_of(name):
        if s == "backbone.rest":
            skip = {id(p) for n in ("layer3", "layer4") for p in getattr(model.backbone, n).parameters()}
            ps = [p for p in model.backbone.parameters() if id(p) not in skip]
        else:
            mod = model
            for part in s.split("."):
                mod = getattr(mod, part)
            ps = list(mod.parameters())
        out += [p for p in ps if p.requires_grad or not trainable_only]
    return out


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
    contiguous buffer; ``reduce(stage)`` - the hook of TrainEngine.backward(on_group_done=) - starts the (asynchronous) all-reduce of
    every bucket whose stages are now all complete, ``finish()`` waits for all of them and turns sums into means."""

    def __init__(self, model, groups=STAGES, process_group=None, optimizer=None, always_reduce=False, comm_dtype=None, timing=False):
        """groups: the buckets, each a stage of STAGES or a coarse group of GROUPS (= all stages under that name); default one bucket per
        stage.  always_reduce: issue the all-reduces even in a one-rank group (a no-op on the values; the single-GPU RCCL test uses it to
        put real RCCL kernels between the backward and the optimizer).
        optimizer: the rdpn6d_amd Ranger that will step these parameters.  If it has already built its flat gradient
        buffer (a step was taken) that buffer is ADOPTED here - each bucket is a contiguous slice of it in any order - so both
        sides keep working on the same memory; an un-built Ranger adopts this buffer at its first step by itself.
        comm_dtype (cfg.SOLVER.ALLREDUCE_DTYPE = "bf16"): the buckets travel as torch.bfloat16 (72.8 instead of 145.6 MB per step for
        ResNet-34: SURVEY.md section 8d) - cast on issue, reduced, cast back into the fp32 buffer in finish(); the mean then carries
        one bf16 rounding per rank-sum step.  None = fp32, bit-identical to a plain all-reduce of the gradients.
        timing: record an event per issued bucket, at finish() entry (= the backward's end) and after its waits - report()."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())
        self.groups = tuple(groups)
        self._stages = {g: stages_of(g) for g in self.groups}
        seen = [s for g in self.groups for s in self._stages[g]]
        if sorted(seen) != sorted(STAGES):
            raise ValueError(f"GradBuckets: buckets {self.groups} must cover every stage of {STAGES} exactly once")
        params = {g: stage_params(model, g) for g in self.groups}
        self.params = params
        n = sum(p.numel() for ps in params.values() for p in ps)
        if n != sum(p.numel() for p in model.parameters() if p.requires_grad):
            raise ValueError("GradBuckets: the stages do not cover the model's trainable parameters exactly")
        ref = next(p for ps in params.values() for p in ps)
        flat = getattr(optimizer, "_flat", None)
        flat = flat["g"] if flat else None
        self.slices = {}
        if flat is not None:
            if flat.numel() != n:
                raise ValueError("GradBuckets: the optimizer's flat gradient buffer does not cover exactly the model's trainable parameters")
            for g in self.groups:
                if not params[g]:
                    self.slices[g] = (0, 0)
                    continue
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
            for g in self.groups:
                b = o
                for p in params[g]:
                    p.grad = self.flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
                self.slices[g] = (b, o)
        self._home = {id(p): (p.grad.data_ptr() - self.flat.data_ptr()) // 4 for ps in params.values() for p in ps}
        self.handles = []
        self._done, self._issued = set(), []
        # gloo has no AVG: reduce with SUM and scale in finish()
        self.avg_op = self.active and dist.get_backend(process_group) == "nccl"
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self._lp = torch.empty(n, dtype=self.comm_dtype, device=ref.device) if (self.comm_dtype is not None and self.active) else None
        self.timing = bool(timing) and self.flat.is_cuda
        self._ev = {}

    @classmethod
    def from_cfg(cls, model, cfg, **kw):
        """buckets as the config asks: cfg.SOLVER.ALLREDUCE_DTYPE "f32" (default) | "bf16", cfg.SOLVER.GRAD_BUCKETS stages | coarse"""
        sol = cfg.get("SOLVER", {})
        dt = str(sol.get("ALLREDUCE_DTYPE", "f32")).lower()
        if dt not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError(f"SOLVER.ALLREDUCE_DTYPE {dt!r}: f32 | bf16")
        kw.setdefault("comm_dtype", torch.bfloat16 if dt.startswith("b") else None)
        kw.setdefault("groups", GROUPS if str(sol.get("GRAD_BUCKETS", "stages")) == "coarse" else STAGES)
        return cls(model, **kw)

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

    def reduce(self, name):
        """`name`: the stage the backward has just completed (what TrainEngine.backward hands to on_group_done), or a bucket / coarse
        group name (= all of its stages).  Every bucket whose stages are now all complete is all-reduced, in bucket order."""
        self._done.update(stages_of(name))
        for g in self.groups:
            if g not in self._issued and all(s in self._done for s in self._stages[g]):
                self._issue(g)

    def _issue(self, group):
        self._issued.append(group)
        self._rehome(group)
        if self.timing:
            self._ev[group] = torch.cuda.Event(enable_timing=True)
            self._ev[group].record()
        if not self.active:
            return
        b, e = self.slices[group]
        if e == b:
            return
        op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
        buf = self.flat[b:e]
        if self._lp is not None:
            buf = self._lp[b:e]
            buf.copy_(self.flat[b:e])
        self.handles.append(dist.all_reduce(buf, op=op, group=self.pg, async_op=True))

    def finish(self):
        """wait for the issued all-reduces (all buckets must have been issued: a forgotten one would leave the ranks' weights apart)"""
        if self.active and len(self._issued) != len(self.groups):
            missing = [g for g in self.groups if g not in self._issued]
            self._done, self._issued = set(), []
            raise RuntimeError(f"GradBuckets.finish: bucket(s) {missing} were never reduced this step")
        if self.timing:
            self._ev["__backward_end__"] = torch.cuda.Event(enable_timing=True)
            self._ev["__backward_end__"].record()
        for h in self.handles:
            h.wait()
        self.handles = []
        if self._lp is not None and self.active:
            for g in self._issued:
                b, e = self.slices[g]
                self.flat[b:e].copy_(self._lp[b:e])
        if self.active and not self.avg_op:
            self.flat.div_(self.world)
        if self.timing:
            self._ev["__reduced__"] = torch.cuda.Event(enable_timing=True)
            self._ev["__reduced__"].record()
        self.last_issue_order = list(self._issued)
        self._done, self._issued = set(), []

    def report(self):
        """(timing=True, after a finish() and a device synchronise) per bucket: MB, and how many ms BEFORE the backward's last kernel
        its all-reduce was issued on the compute stream - the window it has to hide in; `allreduce_exposed_ms` = compute-stream time
        from the backward's end until every all-reduce (and the cast back / the division) had completed."""
        if not self.timing or "__reduced__" not in self._ev:
            return None
        end = self._ev["__backward_end__"]
        out = {"buckets": [{"name": g, "mbytes": round((self.slices[g][1] - self.slices[g][0]) * (2 if self._lp is not None else 4) / 1e6, 1),
                            "issued_ms_before_backward_end": round(self._ev[g].elapsed_time(end), 3)} for g in self.last_issue_order],
               "allreduce_exposed_ms": round(end.elapsed_time(self._ev["__reduced__"]), 3),
               "comm_dtype": str(self.comm_dtype or torch.float32).replace("torch.", ""), "world": self.world, "collectives_issued": self.active}
        return out


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
