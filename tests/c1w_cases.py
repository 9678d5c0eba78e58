"""Shared between tools/oracle/gen_model_golden_w.py (build container, real reference) and the tests: which entries of
each parameter gradient the well-conditioned fixture (tests/golden/model_c1w.npz) stores, and how its weights are made."""
import zlib

import numpy as np

NSAMPLE = 256


def grad_sample_index(name, numel):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([4242, zlib.crc32(name.encode())])))
    return rng.integers(0, numel, size=min(NSAMPLE, numel))


def c1w_state_dict(shapes, bn_npz):
    """seeded trained-like weights + the shipped BatchNorm statistics (numpy)"""
    from rdpn6d_amd import synth

    sd = synth.make_trained_like_state_dict(shapes, seed=1234)
    sd.update({k: bn_npz[k] for k in bn_npz.files})
    return sd
