import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.environ["TAP_INNER"] = "1"
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    import bench_conv as bc
lib = bc.lib
res = {0: [], 1: []}
for rnd in range(6):
    for ti in (1, 0):
        lib.rdpn6d_conv_set_tap_inner(ti)
        with contextlib.redirect_stdout(buf):
            ms = bc.run(*bc.SHAPES[0], reps=20)
        res[ti].append(ms)
for ti in (0, 1):
    v = sorted(res[ti]); fl = 2.0 * bc.B * 64 * 64 * 256 * 9 * 256
    print("tap_inner", ti, "median ms", v[len(v)//2], "TF/s", fl / v[len(v)//2] / 1e9, "min", fl / v[0] / 1e9 , "all", [round(fl/x/1e9,1) for x in res[ti]])
