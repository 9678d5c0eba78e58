"""BUILD-CONTAINER ONLY: golden trajectories from the reference's OWN Ranger class and LR scheduler
(lib/torch_utils/solver/{ranger,lr_scheduler}.py - pure torch, imported from /root/reference).

  python tools/oracle/gen_ranger_golden.py     # writes tests/golden/ranger_golden.npz
Seeded small tensors (conv 4-D, fc 2-D, norm 1-D), 14 steps: covers the un-rectified start (N_sma <= 5),
the rectified regime and two lookahead syncs (steps 6 and 12)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))
import ref_stubs  # noqa: E402

ref_stubs.install()
from lib.torch_utils.solver.ranger import Ranger as RefRanger  # noqa: E402
from lib.torch_utils.solver.lr_scheduler import flat_and_anneal_lr_scheduler as ref_sched  # noqa: E402
from tests.ranger_cases import SHAPES, make_params, make_grads  # noqa: E402


def main():
    out = {}
    ps = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in make_params()]
    opt = RefRanger([{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 3e-2}], lr=1e-2, weight_decay=0)
    for step in range(14):
        for p, g in zip(ps, make_grads(step)):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        for i, p in enumerate(ps):
            out[f"s{step}_p{i}"] = p.detach().numpy().copy()
    # LR factors
    dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    for name, kw in (("cos", dict(total_iters=1000, warmup_iters=100, warmup_factor=0.001, anneal_point=0.72, anneal_method="cosine")),
                     ("lin", dict(total_iters=500, warmup_iters=0, anneal_point=0.5, anneal_method="linear", target_lr_factor=0.1)),
                     ("poly", dict(total_iters=400, warmup_iters=50, warmup_factor=0.1, anneal_point=0.6, anneal_method="poly", poly_power=0.9))):
        s = ref_sched(dummy, **kw)
        lam = s.lr_lambdas[0]
        out["lr_" + name] = np.array([lam(x) for x in range(kw["total_iters"] + 1)], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ranger_golden.npz"), **out)
    print("wrote ranger_golden.npz", len(out))


if __name__ == "__main__":
    main()
