"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path (the product is the fused HIP step
in rdpn6d_amd/ranger.py + rdpn6d_amd/csrc/ranger.hip).

torch (CPU or GPU) restatement of the reference's Ranger = RAdam + Lookahead + gradient centralisation
(/root/reference/lib/torch_utils/solver/ranger.py:100-200: lr 1e-3, alpha 0.5, k 6, N_sma_threshhold 5,
betas (0.95, 0.999), eps 1e-5, GC on conv + fc) and of flat_and_anneal_lr_scheduler's factor
(lib/torch_utils/solver/lr_scheduler.py:177-263).
Parity is PINNED: tests/golden/ranger_golden.npz holds parameter trajectories produced by the reference's own
Ranger class and LR factors from its own scheduler (tools/oracle/gen_ranger_golden.py); tests/test_ranger.py
checks this file against them.
"""
import math

import torch
from torch.optim.optimizer import Optimizer


class Ranger(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False):
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

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            grads = []
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                    st["slow_buffer"] = p.detach().clone()
                g = p.grad.float()
                if self.use_gc and g.dim() > self.gc_min_dim:
                    g = g - g.mean(dim=tuple(range(1, g.dim())), keepdim=True)
                grads.append(g)
                st["step"] += 1
            step = self.state[ps[0]]["step"]  # all tensors of a group step together
            m = [self.state[p]["exp_avg"] for p in ps]
            v = [self.state[p]["exp_avg_sq"] for p in ps]
            torch._foreach_mul_(v, beta2)
            torch._foreach_addcmul_(v, grads, grads, value=1 - beta2)
            torch._foreach_mul_(m, beta1)
            torch._foreach_add_(m, grads, alpha=1 - beta1)
            beta2_t = beta2 ** step
            n_max = 2 / (1 - beta2) - 1
            n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
            if group["weight_decay"] != 0:
                torch._foreach_mul_(ps, 1 - group["weight_decay"] * group["lr"])
            if n_sma > group["N_sma_threshhold"]:
                step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max /
                                      (n_max - 2)) / (1 - beta1 ** step)
                denom = torch._foreach_sqrt(v)
                torch._foreach_add_(denom, group["eps"])
                torch._foreach_addcdiv_(ps, m, denom, value=-step_size * group["lr"])
            else:
                torch._foreach_add_(ps, m, alpha=-group["lr"] / (1 - beta1 ** step))
            if step % group["k"] == 0:
                slow = [self.state[p]["slow_buffer"] for p in ps]
                torch._foreach_lerp_(slow, ps, group["alpha"])
                torch._foreach_copy_(ps, slow)
        return None


def flat_and_anneal_factor(x, total_iters, warmup_iters=0, warmup_factor=0.1, warmup_method="linear", anneal_point=0.72,
                           anneal_method="cosine", target_lr_factor=0.0, poly_power=1.0):
    """lr factor of lr_scheduler.py:218-258 for the methods the configs use (cosine | linear | poly | none)."""
    anneal_start = anneal_point * total_iters
    if x < warmup_iters:
        if warmup_method == "linear":
            a = float(x) / warmup_iters
            return warmup_factor * (1 - a) + a
        return warmup_factor
    if x >= anneal_start:
        if anneal_method == "cosine":
            return target_lr_factor + 0.5 * (1 - target_lr_factor) * (1 + math.cos(math.pi * ((float(x) - anneal_start) / (total_iters - anneal_start))))
        if anneal_method == "linear":
            return target_lr_factor + (1 - target_lr_factor) * (total_iters - float(x)) / (total_iters - anneal_start)
        if anneal_method == "poly":
            return target_lr_factor + (1 - target_lr_factor) * ((total_iters - float(x)) / (total_iters - anneal_start)) ** poly_power
        return 1
    return 1
