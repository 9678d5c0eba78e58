"""flat_and_anneal LR schedule (host logic) with the reference's signature
(lib/torch_utils/solver/lr_scheduler.py:177-263; used by every RGB-D config through
SOLVER.LR_SCHEDULER_NAME="flat_and_anneal", ANNEAL_METHOD="cosine", ANNEAL_POINT=0.72, WARMUP 1000 x 0.001)."""
from bisect import bisect_right
from math import cos, pi

import torch


def flat_and_anneal_lr_scheduler(optimizer, total_iters, warmup_iters=0, warmup_factor=0.1, warmup_method="linear",
                                 anneal_point=0.72, anneal_method="cosine", target_lr_factor=0, poly_power=1.0, step_gamma=0.1,
                                 steps=(2 / 3.0, 8 / 9.0)):
    if warmup_method not in ("constant", "linear"):
        raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
    if anneal_method not in ("cosine", "linear", "poly", "exp", "step", "none"):
        raise ValueError("Only 'cosine', 'linear', 'poly', 'exp', 'step' or 'none' anneal_method accepted, got {}".format(anneal_method))
    if anneal_method == "step":
        if any(s < warmup_iters / total_iters or s > 1 for s in steps):
            raise ValueError("error in steps: {}".format(steps))
        if list(steps) != sorted(steps):
            raise ValueError("steps {} is not in ascending order.".format(steps))
        anneal_start = steps[0] * total_iters
    else:
        if anneal_point > 1 or anneal_point < 0:
            raise ValueError("anneal_point should be in [0,1], got {}".format(anneal_point))
        anneal_start = anneal_point * total_iters

    def f(x):
        if x < warmup_iters:
            if warmup_method == "linear":
                a = float(x) / warmup_iters
                return warmup_factor * (1 - a) + a
            return warmup_factor
        if x >= anneal_start:
            if anneal_method == "step":
                return step_gamma ** bisect_right([s * total_iters for s in steps], float(x))
            if anneal_method == "none":
                return 1
            frac = (float(x) - anneal_start) / (total_iters - anneal_start)  # progress through the annealing stretch
            if anneal_method == "cosine":
                return target_lr_factor + 0.5 * (1 - target_lr_factor) * (1 + cos(pi * frac))
            if anneal_method == "linear":
                return target_lr_factor + (1 - target_lr_factor) * (1 - frac)
            if anneal_method == "poly":
                return target_lr_factor + (1 - target_lr_factor) * (1 - frac) ** poly_power
            return max(target_lr_factor, 5e-3) ** frac  # "exp"
        return 1

    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)
