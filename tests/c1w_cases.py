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


# tests/golden/model_c1w_seeds.npz (tools/oracle/gen_model_golden_seeds.py): the same fixture on EIGHT CONSECUTIVE input seeds, no search.
SEEDS = tuple(range(8))
# THE TIE RULE: a pixel is in the tie set iff the reference's own top-2 region-logit gap is below TIE_GAP (two logits moving by the
# map tolerance 1e-4 in opposite directions can swap) or the reference flips it against itself (1 vs 8 threads, fp32 vs float64)
TIE_GAP = 2e-4


def tie_set(gold, s):
    """bool (4, 64, 64): the recorded tie set of seed s"""
    shape = gold[f"s{s}_top2_gap"].shape
    n = int(np.prod(shape))
    f18 = np.unpackbits(gold[f"s{s}_flip_1v8"])[:n].reshape(shape).astype(bool)
    f64 = np.unpackbits(gold[f"s{s}_flip_fp64"])[:n].reshape(shape).astype(bool)
    return (gold[f"s{s}_top2_gap"] < float(gold["tie_gap"])) | f18 | f64
