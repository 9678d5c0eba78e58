"""The C restatement of the reference FPS (oracle/fps_oracle.c) against golden index vectors
produced by the reference's own .cpp (tools/oracle/gen_fps_golden.py), and against that library
directly when oracle/_ref is present."""
import ctypes
import os

import numpy as np
import pytest

from tests.fps_cases import fps_cases, make_cloud

P = ctypes.c_void_p


def _oracle(lib, pts, sn, mode):
    idx = np.full(sn, -7, dtype=np.int32)
    if mode == "center":
        lib.oracle_fps_init_center(pts.ctypes.data_as(P), idx.ctypes.data_as(P), len(pts), sn)
    else:
        lib.oracle_fps_from_start(pts.ctypes.data_as(P), idx.ctypes.data_as(P), len(pts), sn, int(mode))
    return idx


def test_oracle_matches_reference_golden(oracle_lib, golden_dir):
    gold = np.load(os.path.join(golden_dir, "fps_golden.npz"))
    cases = fps_cases()
    assert len(cases) == len(gold.files)
    for name, kind, n, sn, seed, mode in cases:
        idx = _oracle(oracle_lib, make_cloud(kind, n, seed), sn, mode)
        assert np.array_equal(idx, gold[name]), name


def test_oracle_matches_reference_library_if_present(oracle_lib):
    ref = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libfps_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (reference absent)")
    lib = ctypes.CDLL(ref)
    rng = np.random.default_rng(0)
    for n, sn in ((257, 16), (4096, 33), (20000, 64)):
        pts = np.ascontiguousarray(rng.standard_normal((n, 3)), dtype=np.float32)
        want = np.zeros(sn, dtype=np.int32)
        lib.farthest_point_sampling_init_center(pts.ctypes.data_as(P), want.ctypes.data_as(P), n, sn)
        assert np.array_equal(_oracle(oracle_lib, pts, sn, "center"), want)


def test_edge_semantics(oracle_lib):
    # all-identical points: every later pick falls back to index 0 (strict '>' from max_d=0)
    pts = make_cloud("same", 100, 3)
    assert _oracle(oracle_lib, pts, 8, "7").tolist() == [7, 0, 0, 0, 0, 0, 0, 0]
    # sn > N: once every point is taken the arg-max returns 0 for ever
    pts = make_cloud("gauss", 16, 5)
    idx = _oracle(oracle_lib, pts, 40, "center")
    assert sorted(idx[:16].tolist()) == list(range(16)) and (idx[16:] == 0).all()
