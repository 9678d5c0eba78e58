"""Conv-stack fraction of the fp32-accurate ceiling (833 TFLOP/s = 2 500 dense fp16 MFMA / 3 partial products), computed from a
`rocprofv3 --kernel-trace --stats` summary of `bench.py` the way VERDICT r3 did: executed conv GFLOP per step / time in conv kernels.

    python tools/conv_stack_fraction.py profiles/r4_a_kernel_stats.csv [executed GFLOP per crop = 40.48] [batch = 64]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gflop_crop = float(sys.argv[2]) if len(sys.argv) > 2 else 40.48  # 44.10 of the reference's order - 3.62 the exact rewrites do not execute
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
steps = next(int(r["Calls"]) for r in rows if "stem_pool_h2" in r["Name"])  # one fused stem launch per step
conv = [r for r in rows if any(k in r["Name"] for k in ("conv_h2", "conv_igemm", "conv_x3", "stem_pool_h2"))]
tot_all = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
tot = sum(float(r["TotalDurationNs"]) for r in conv) / steps / 1e6
print(f"{steps} steps profiled; kernel time per step {tot_all:.3f} ms, of which convolution kernels {tot:.3f} ms:")
for r in sorted(conv, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print(f"  {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms/step  {int(r['Calls']) / steps:5.1f} launches x {float(r['AverageNs']) / 1e3:7.1f} us  {n}")
tf = gflop_crop * B / tot
print(f"conv stack: {gflop_crop * B:.0f} GFLOP / {tot:.3f} ms = {tf:.0f} TFLOP/s = {tf / 833.3:.3f} of 833 TFLOP/s")
