"""BUILD-CONTAINER ONLY: golden index vectors for farthest point sampling, produced by the
reference's OWN .cpp compiled in place (oracle/Makefile -> oracle/_ref/libfps_ref.so).

  make -C oracle && python tools/oracle/gen_fps_golden.py     # writes tests/golden/fps_golden.npz

Cases: N in {64,1000,5000,50000} x sn in {8,32,64,256} x
{gaussian, unit-sphere surface, 10^3 lattice (exact ties), all-identical, sn>N}
x {init_center, start s in {0,7,N-1}}.  The random-start entry point of the reference
(farthest_point_sampling.cpp:93-94) is pinned with an LD_PRELOAD rand() shim in a subprocess.
Only seeds + int32 indices are stored; tests regenerate the clouds with fps_cases().
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from tests.fps_cases import fps_cases, make_cloud  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "libfps_ref.so")
SHIM = os.path.join(ROOT, "oracle", "_ref", "librand_shim.so")

CHILD = r"""
import ctypes, sys, json, numpy as np
sys.path.insert(0, %r)
from tests.fps_cases import make_cloud
lib = ctypes.CDLL(%r)
kind, n, sn, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pts = make_cloud(kind, n, seed)
idx = np.zeros(sn, dtype=np.int32)
lib.farthest_point_sampling(pts.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p), n, sn)
print(json.dumps(idx.tolist()))
"""


def main():
    lib = ctypes.CDLL(REF)
    out = {}
    for name, kind, n, sn, seed, mode in fps_cases():
        pts = make_cloud(kind, n, seed)
        if mode == "center":
            idx = np.zeros(sn, dtype=np.int32)
            lib.farthest_point_sampling_init_center(
                pts.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p), n, sn)
        else:
            start = int(mode)
            env = dict(os.environ, LD_PRELOAD=SHIM, FAKE_RAND=str(start))
            r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, REF), kind, str(n), str(sn), str(seed)],
                               env=env, capture_output=True, text=True, check=True)
            idx = np.array(json.loads(r.stdout.strip().splitlines()[-1]), dtype=np.int32)
        out[name] = idx
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fps_golden.npz"), **out)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
