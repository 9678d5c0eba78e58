"""Ranger + flat_and_anneal: the torch restatement (oracle) and the host scheduler against golden trajectories produced
by the reference's own classes; on the GPU the fused HIP step against the same golden file."""
import os

import numpy as np
import pytest
import torch

from tests.ranger_cases import SHAPES, make_grads, make_params


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ranger_golden.npz"))


def _run(opt_cls, device, **kw):
    ps = [torch.nn.Parameter(torch.from_numpy(a.copy()).to(device)) for a in make_params()]
    opt = opt_cls([{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 3e-2}], lr=1e-2, weight_decay=0, **kw)
    traj = []
    for step in range(14):
        for p, g in zip(ps, make_grads(step)):
            gg = torch.from_numpy(g.copy()).to(device)
            if p.grad is None:
                p.grad = gg
            else:
                p.grad.copy_(gg)
        opt.step()
        traj.append([p.detach().cpu().numpy().copy() for p in ps])
    return traj, opt, ps


def _check(traj, gold, tol):
    for step, row in enumerate(traj):
        for i, a in enumerate(row):
            ref = gold[f"s{step}_p{i}"]
            err = np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-6)
            assert err < tol, (step, i, err)


def test_oracle_ranger_matches_reference_golden(gold):
    from oracle.ranger_oracle import Ranger

    traj, _, _ = _run(Ranger, "cpu")
    _check(traj, gold, 2e-6)


def test_scheduler_matches_reference_golden(gold):
    from oracle.ranger_oracle import flat_and_anneal_factor
    from rdpn6d_amd.lr_scheduler import flat_and_anneal_lr_scheduler

    cases = {"cos": dict(total_iters=1000, warmup_iters=100, warmup_factor=0.001, anneal_point=0.72, anneal_method="cosine"),
             "lin": dict(total_iters=500, warmup_iters=0, anneal_point=0.5, anneal_method="linear", target_lr_factor=0.1),
             "poly": dict(total_iters=400, warmup_iters=50, warmup_factor=0.1, anneal_point=0.6, anneal_method="poly", poly_power=0.9)}
    dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    for name, kw in cases.items():
        lam = flat_and_anneal_lr_scheduler(dummy, **kw).lr_lambdas[0]
        got = np.array([lam(x) for x in range(kw["total_iters"] + 1)])
        assert np.abs(got - gold["lr_" + name]).max() < 1e-12, name
        got_o = np.array([flat_and_anneal_factor(x, **kw) for x in range(kw["total_iters"] + 1)])
        assert np.abs(got_o - gold["lr_" + name]).max() < 1e-12, name
    with pytest.raises(ValueError):
        flat_and_anneal_lr_scheduler(dummy, 100, anneal_method="nope")


def test_product_ranger_refuses_cpu():
    from rdpn6d_amd.ranger import Ranger

    p = torch.nn.Parameter(torch.zeros(3, 3))
    p.grad = torch.ones(3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Ranger([p]).step()
    with pytest.raises(ValueError):
        Ranger([p], alpha=2.0)


@pytest.mark.gpu
def test_hip_ranger_matches_reference_golden(gold):
    from rdpn6d_amd.ranger import Ranger

    traj, opt, ps = _run(Ranger, "cuda:0")
    _check(traj, gold, 3e-6)
    # state layout of the reference (per-parameter step / exp_avg / exp_avg_sq / slow_buffer), views of flat buffers
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq", "slow_buffer"} and sd["state"][0]["step"] == 14
    assert all(p.grad.data_ptr() >= opt.flat_grad.data_ptr() for p in ps)
    # replacing p.grad (zero_grad(set_to_none=True) + new backward) is tolerated
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    assert all(torch.isfinite(p).all() for p in ps)


@pytest.mark.gpu
def test_hip_ranger_step_under_a_loss_scale(gold):
    """Ranger.step(grad_scale=S, skip_if_nonfinite=True): what GradScaler.unscale_ + GradScaler.step do around the reference's optimizer
    (engine.py:302-309) inside the optimizer's own launches.  With S a power of two the trajectory on S-times-scaled gradients is the
    un-scaled one BIT FOR BIT (so the reference golden holds); a step with an Inf or a NaN anywhere in the flat gradient changes no
    parameter, no state and not the step count, and the next clean step continues as if it had not happened."""
    from rdpn6d_amd.ranger import Ranger

    dev, S = "cuda:0", 4096.0
    plain, _, _ = _run(Ranger, dev)
    ps = [torch.nn.Parameter(torch.from_numpy(a.copy()).to(dev)) for a in make_params()]
    opt = Ranger([{"params": ps[:2], "lr": 1e-2}, {"params": ps[2:], "lr": 3e-2}], lr=1e-2, weight_decay=0)
    for step in range(14):
        for poison in ((float("inf"), float("nan"))[step % 2:step % 2 + 1] if step in (3, 8) else ()) + (None,):
            for p, g in zip(ps, make_grads(step)):
                gg = torch.from_numpy(g.copy()).to(dev) * S
                if p.grad is None:
                    p.grad = gg
                else:
                    p.grad.copy_(gg)
            if poison is not None:
                ps[4].grad.view(-1)[4321] = poison  # one element of 45 000
                before = [p.detach().clone() for p in ps]
                state = [opt.state[p]["exp_avg"].clone() for p in ps] if step else None
            opt.step(grad_scale=S, skip_if_nonfinite=True)
            if poison is not None:
                assert opt.found_inf() and opt.found_inf()  # (asking twice rewinds the step count once)
                assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps))
                if state is not None:
                    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(state, ps))
                assert opt.state[ps[0]]["step"] == step
            else:
                assert not opt.found_inf()
        for a, p in zip(plain[step], ps):
            assert np.array_equal(a, p.detach().cpu().numpy()), step
    _check([[p.detach().cpu().numpy() for p in ps]], {f"s0_p{i}": gold[f"s13_p{i}"] for i in range(len(ps))}, 3e-6)


@pytest.mark.gpu
def test_hip_ranger_row_means_inside_the_update_kernel(monkeypatch):
    """Gradient centralisation (ranger.py:146-148): a work item that IS a whole row takes the row mean inside the update kernel, with
    gc_row_mean_kernel's own summation; longer rows keep the separate pass.  Same bits either way: the trajectory with every row split
    (work items of 1 024 elements, so the 5 000-element rows go through the separate kernel) equals the default's."""
    import rdpn6d_amd.ranger as rg

    traj, opt, _ = _run(rg.Ranger, "cuda:0")
    assert opt._nrows == 0  # every centralised row of the fixture is one work item
    monkeypatch.setattr(rg, "_CHUNK", 1024)
    traj2, opt2, _ = _run(rg.Ranger, "cuda:0")
    assert opt2._nrows == 9  # the (9, 5000) tensor's rows
    for a, b in zip(traj, traj2):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
