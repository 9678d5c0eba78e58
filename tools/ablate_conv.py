import os, sys, io, contextlib
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.environ["TAP_INNER"] = "1"
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    import bench_conv as bc
lib = bc.lib
fl = 2.0 * bc.B * 64 * 64 * 256 * 9 * 256
for name, v in (("full", 1), ("no global loads", 1 | 2), ("no loads, no lds stores", 1 | 2 | 4), ("no loads/stores/barrier", 1 | 2 | 4 | 8), ("mfma only (no frag reads either)", 1 | 2 | 4 | 8 | 16), ("no barrier only", 1 | 8), ("full again", 1)):
    lib.rdpn6d_conv_set_tap_inner(v)
    with contextlib.redirect_stdout(buf):
        ms = min(bc.run(*bc.SHAPES[0], reps=10) for _ in range(3))
    print(f"{name:40s} {fl/ms/1e9:7.1f} TF/s ({fl/ms/1e9/157.3*100:5.1f}%)")
