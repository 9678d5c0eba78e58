"""seeded parameter / gradient tensors shared by the Ranger golden generator and the tests"""
import numpy as np

SHAPES = [(8, 4, 3, 3), (5, 7), (16,), (6, 3, 1, 1), (9, 5000)]


def _rng(*salt):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([77] + list(salt))))


def make_params():
    return [(_rng(1, i).standard_normal(s) * 0.5).astype(np.float32) for i, s in enumerate(SHAPES)]


def make_grads(step):
    return [(_rng(2, step, i).standard_normal(s) * (0.1 + 0.05 * i) + 0.02).astype(np.float32) for i, s in enumerate(SHAPES)]
