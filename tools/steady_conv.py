import os, sys, io, contextlib
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.environ["TAP_INNER"] = "1"
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    import bench_conv as bc
for idx in (0, 1, 2, 3, 4, 5):
    s = bc.SHAPES[idx]
    with contextlib.redirect_stdout(buf):
        ms = sorted(bc.run(*s, reps=20) for _ in range(5))
    name, H, Cin, Cout, k, st = s
    taps = 4 if k == 2 else k * k
    fl = 2.0 * bc.B * H * H * Cout * taps * Cin
    print(f"{name:38s} median {fl/ms[2]/1e9:7.1f} TF/s ({fl/ms[2]/1e9/157.3*100:5.1f}%)  best {fl/ms[0]/1e9:7.1f}")
