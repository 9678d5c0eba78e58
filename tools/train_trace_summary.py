"""Per-launch view of a rocprofv3 --kernel-trace CSV: one row per (kernel, grid) bucket with its call count and mean / min duration, sorted by
total time - the averages of --stats hide that one kernel name covers launches from 4 us to 40 us (BatchNorm passes over 2 MB and 67 MB maps).
usage: python tools/train_trace_summary.py <kernel_trace.csv> [steps] [top]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 150
b = defaultdict(list)
for r in csv.DictReader(open(path)):
    name = re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = name.split("(")[0][:70]
    grid = tuple(int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = int(r["Workgroup_Size_X"])
    b[(name, grid[0] // max(wg, 1), grid[1], grid[2])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(b.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in b.values())
print(f"total kernel time {tot / 1e3:.2f} ms over {steps} steps = {tot / steps / 1e3:.3f} ms per step; {len(rows)} (kernel, grid) buckets")
print(f"{'kernel':70s} {'workgroups':>16s} {'calls/step':>10s} {'mean us':>8s} {'min us':>8s} {'us/step':>8s}")
for (name, gx, gy, gz), v in rows[:top]:
    print(f"{name:70s} {f'{gx}x{gy}x{gz}':>16s} {len(v) / steps:10.2f} {sum(v) / len(v):8.1f} {min(v):8.1f} {sum(v) / steps:8.1f}")
