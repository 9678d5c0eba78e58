"""Ranger (RAdam + Lookahead + gradient centralisation) as ONE fused HIP step over flat HBM buffers.

Same update rule, hyper-parameter names and defaults as the reference's optimizer
(lib/torch_utils/solver/ranger.py:27-200: lr 1e-3, alpha 0.5, k 6, N_sma_threshhold 5, betas (0.95, 0.999),
eps 1e-5, gradient centralisation on conv + fc weights), so ``SOLVER.OPTIMIZER_CFG=dict(type="Ranger", ...)``
works unchanged; the 164-tensor Python loop becomes two kernel launches per parameter group
(rdpn6d_amd/csrc/ranger.hip).  Parameters, gradients, exp_avg, exp_avg_sq and the lookahead slow weights each
live in one contiguous buffer; ``param.data`` / ``param.grad`` / ``state[p][...]`` are views into them, so
``state_dict()`` keeps the reference's per-parameter layout (step, exp_avg, exp_avg_sq, slow_buffer).

There is no CPU path: parameters must live on the GPU (the torch restatement used by the tests is
oracle/ranger_oracle.py).  SURVEY.md section 8f rank 2.
"""
import ctypes
import math

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from . import _lib

_CHUNK = 4096  # elements per work item (one 256-thread workgroup)
_ROW_HERE = 0x40000000  # RangerWork.row: the item is a whole gradient-centralisation row (csrc/ranger.hip)


class Ranger(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False, flat_grad=None):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        super().__init__(params, dict(lr=lr, alpha=alpha, k=k, betas=betas, N_sma_threshhold=N_sma_threshhold, eps=eps,
                                      weight_decay=weight_decay))
        self.use_gc = use_gc
        self.gc_min_dim = 3 if gc_conv_only else 1
        self._flat = None
        self._ext_flat_grad = flat_grad
        self._step = 0

    # ------------------------------------------------------------------ flat storage
    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps:
            raise ValueError("Ranger: no trainable parameters")
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("rdpn6d_amd.Ranger runs its fused step on the GPU only (no CPU fallback); got CPU parameters")
        n = sum(p.numel() for p in ps)
        f32 = dict(dtype=torch.float32, device=dev)
        fp = torch.empty(n, **f32)
        # Gradients.  If every p.grad already is a view into ONE foreign flat fp32 buffer that the parameters tile exactly
        # (parallel.GradBuckets, in whatever group order it chose), that buffer IS adopted and its layout becomes the layout
        # of all five flat buffers - the all-reduce views stay valid.  The work tables below address parameters by offset,
        # so no particular order is needed.  Otherwise a fresh buffer in param_groups order is made.
        fg = self._ext_flat_grad
        if fg is None and all(p.grad is not None for p in ps):
            from .parallel import flat_grad_storage

            fg = flat_grad_storage([p.grad for p in ps])  # (by storage: survives autograd's AccumulateGrad re-adopting the views)
        offs, adopt = {}, False
        if fg is not None:
            adopt = fg.dim() == 1 and fg.numel() == n and fg.dtype == torch.float32 and fg.is_contiguous() and fg.device == dev
            spans = []
            for p in ps:
                if not adopt:
                    break
                byte = (p.grad.data_ptr() - fg.data_ptr()) if p.grad is not None else -1
                if byte < 0 or byte % 4 or byte // 4 + p.numel() > n or not p.grad.is_contiguous():
                    adopt = False
                    break
                offs[id(p)] = byte // 4
                spans.append((byte // 4, p.numel()))
            if adopt:  # exact tiling of [0, n): no overlap, no hole
                spans.sort()
                adopt = all(spans[i][0] + spans[i][1] == (spans[i + 1][0] if i + 1 < len(spans) else n) for i in range(len(spans))) \
                    and spans[0][0] == 0
            if not adopt and self._ext_flat_grad is not None:
                raise ValueError("Ranger(flat_grad=...): the parameters' .grad tensors are not views that tile this buffer exactly; "
                                 "build parallel.GradBuckets(model) first (it points every param.grad into its flat buffer) or "
                                 "drop the argument")
        if not adopt:
            fg = torch.zeros(n, **f32)
            offs, o = {}, 0
            for p in ps:
                offs[id(p)] = o
                o += p.numel()
        fm, fv, fs = torch.zeros(n, **f32), torch.zeros(n, **f32), torch.empty(n, **f32)
        for p in ps:
            o, k = offs[id(p)], p.numel()
            fp[o:o + k].copy_(p.detach().reshape(-1))
            p.data = fp[o:o + k].view_as(p)
            if not adopt:
                if p.grad is not None:
                    fg[o:o + k].copy_(p.grad.reshape(-1))
                p.grad = fg[o:o + k].view_as(p)
            st = self.state[p]
            st["step"] = self._step
            st["exp_avg"], st["exp_avg_sq"] = fm[o:o + k].view_as(p), fv[o:o + k].view_as(p)
            st["slow_buffer"] = fs[o:o + k].view_as(p)
        fs.copy_(fp)
        # work tables per group
        self._groups = []
        row_off, row_len = [], []
        for g in self.param_groups:
            work = []
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                o, k = offs[id(p)], p.numel()
                if self.use_gc and p.dim() > self.gc_min_dim:
                    rl = k // p.shape[0]
                    for r in range(p.shape[0]):
                        if rl <= 2 * _CHUNK:  # the whole row is one work item: the update kernel takes the row mean itself
                            work.append((o + r * rl, rl, _ROW_HERE))
                            continue
                        ridx = len(row_off)
                        row_off.append(o + r * rl)
                        row_len.append(rl)
                        for c in range(0, rl, _CHUNK):
                            work.append((o + r * rl + c, min(_CHUNK, rl - c), ridx))
                else:
                    for c in range(0, k, _CHUNK):
                        work.append((o + c, min(_CHUNK, k - c), -1))
            wt = np.zeros(len(work), dtype=np.dtype([("off", "<i8"), ("len", "<i4"), ("row", "<i4")]))
            for i, (a, b, c) in enumerate(work):
                wt[i] = (a, b, c)
            self._groups.append(torch.from_numpy(wt.view(np.uint8).copy()).to(dev))
        self._nwork = [g.numel() // 16 for g in self._groups]
        self._row_off = torch.tensor(row_off or [0], dtype=torch.int64, device=dev)
        self._row_len = torch.tensor(row_len or [1], dtype=torch.int32, device=dev)
        self._nrows = len(row_off)
        self._row_mean = torch.zeros(max(self._nrows, 1), **f32)
        self._flat = dict(p=fp, g=fg, m=fm, v=fv, s=fs)
        self._params = ps

    @property
    def flat_grad(self):
        if self._flat is None:
            self._build()
        return self._flat["g"]

    def zero_grad(self, set_to_none=False):
        """Zeroes IN PLACE, whatever `set_to_none` says (the reference loop calls zero_grad(set_to_none=True), engine.py:304):
        param.grad tensors are views into a flat buffer - this optimizer's, or parallel.GradBuckets' all-reduce buffer - and
        dropping them would silently detach the gradients from the buffer that is reduced / stepped."""
        if self._flat is not None:
            self._flat["g"].zero_()
            return
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if grads:
            torch._foreach_zero_(grads)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, skip_if_nonfinite=False):
        """grad_scale / skip_if_nonfinite (round 5): the step under a loss scale as the reference runs it (GradScaler.unscale_ +
        GradScaler.step, engine.py:302-309) without the separate passes - every gradient is read as grad / grad_scale (the buffer
        keeps the scaled values), and with skip_if_nonfinite one pass sets a device flag for NaN / Inf anywhere in the flat gradient
        and the update kernels skip the whole step on it; found_inf() reads the flag (the one host read GradScaler.step has).  A guarded
        step whose flag nobody read is resolved at the start of the next step(), whatever its kind: a skipped step never counts."""
        if self._flat is None:
            self._build()
        F = self._flat
        for p in self._params:  # a caller may have replaced p.grad (e.g. zero_grad(set_to_none=True) + backward)
            o = p.data_ptr() - F["p"].data_ptr()
            if p.grad is None:
                raise RuntimeError("Ranger.step: a parameter has no gradient")
            if p.grad.data_ptr() != F["g"].data_ptr() + o:
                F["g"][o // 4:o // 4 + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = F["g"][o // 4:o // 4 + p.numel()].view_as(p)
        if getattr(self, "_found_inf_pending", False):
            self.found_inf()  # the previous guarded step was never asked about: resolve it now, so a skipped step is rewound exactly once
        self._step += 1
        step = self._step
        from .gdrn import bump_weights_epoch

        bump_weights_epoch(self._params)  # the kernel writes THESE parameters through raw pointers: their model's packed inference copies are stale
        lib = _lib.load()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        first = True
        scaled = float(grad_scale) != 1.0 or skip_if_nonfinite
        if skip_if_nonfinite:
            if getattr(self, "_found_inf", None) is None:
                self._found_inf = torch.zeros(1, dtype=torch.int32, device=F["g"].device)
            _lib.check(lib.rdpn6d_grad_nonfinite_f32(P(F["g"]), F["g"].numel(), P(self._found_inf), st), "grad_nonfinite")
            self._found_inf_pending, self._found_inf_last = True, False
        for gi, g in enumerate(self.param_groups):
            if self._nwork[gi] == 0:
                continue
            beta1, beta2 = g["betas"]
            beta2_t = beta2 ** step
            n_max = 2 / (1 - beta2) - 1
            n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
            rect = n_sma > g["N_sma_threshhold"]
            if rect:
                step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (1 - beta1 ** step)
            else:
                step_size = 1.0 / (1 - beta1 ** step)
            if scaled:
                _lib.check(lib.rdpn6d_ranger_step_scaled_f32(
                    P(F["p"]), P(F["g"]), P(F["m"]), P(F["v"]), P(F["s"]), P(self._groups[gi]), self._nwork[gi], P(self._row_off),
                    P(self._row_len), self._nrows if first else 0, P(self._row_mean), beta1, beta2, g["eps"], -step_size * g["lr"],
                    g["weight_decay"] * g["lr"], 1 if rect else 0, 1 if step % g["k"] == 0 else 0, g["alpha"], 1.0 / float(grad_scale),
                    P(self._found_inf) if skip_if_nonfinite else None, st), "ranger_step")
            else:
                _lib.check(lib.rdpn6d_ranger_step_f32(
                    P(F["p"]), P(F["g"]), P(F["m"]), P(F["v"]), P(F["s"]), P(self._groups[gi]), self._nwork[gi], P(self._row_off),
                    P(self._row_len), self._nrows if first else 0, P(self._row_mean), beta1, beta2, g["eps"], -step_size * g["lr"],
                    g["weight_decay"] * g["lr"], 1 if rect else 0, 1 if step % g["k"] == 0 else 0, g["alpha"], st), "ranger_step")
            first = False  # the row means cover all groups and are computed once
        for p in self._params:
            self.state[p]["step"] = step
        return None

    def found_inf(self):
        """True when the last step(skip_if_nonfinite=True) found a NaN / Inf gradient and left parameters and state untouched (one host
        read; the step counter has advanced all the same - rewound here, as a skipped GradScaler step does not count)."""
        if getattr(self, "_found_inf", None) is None:  # no step(skip_if_nonfinite=True) yet
            return False
        if not getattr(self, "_found_inf_pending", False):  # asked again about the same step: the answer it gave (no second rewind)
            return self._found_inf_last
        bad = bool(int(self._found_inf.item()))
        if bad:
            self._step -= 1
            for p in self._params:
                self.state[p]["step"] = self._step
        self._found_inf_pending, self._found_inf_last = False, bad
        return bad

    def load_state_dict(self, state_dict):
        """copy a reference-format optimizer state (per-parameter step / exp_avg / exp_avg_sq / slow_buffer) into the flat buffers"""
        if self._flat is None:
            self._build()
        views = {id(p): dict(self.state[p]) for p in self._params}
        super().load_state_dict(state_dict)
        for p in self._params:
            st, v = self.state[p], views[id(p)]
            for k in ("exp_avg", "exp_avg_sq", "slow_buffer"):
                v[k].copy_(st[k])
                st[k] = v[k]
            self._step = int(st.get("step", self._step))
