# per-call kernel durations of one bench.py line, grouped by (kernel, grid size):  bash tools/debug/run_ktrace.sh <outdir> <filter-regex> <bench.py args...>
R=$(pwd); OUT=$R/gpurun_out/$1; FILT=$2; shift; shift; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt2
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python3 $R/bench.py "$@" --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench.json 2>$OUT/err.log
f=$(ls /tmp/kt2/*/*kernel_trace.csv | head -1)
python3 - "$f" "$FILT" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not pat.search(n): continue
    key = (n.replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Grid_Size_X"], r.get("Grid_Size_Y", ""), r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0, 1e9])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for k, (c, t, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/c:9.1f} us avg  {mn:9.1f} min  x{c:4d}  total {t/1e3:8.3f} ms  grid {k[1]:>8}x{k[2]:>5} wg {k[3]:>4}  {k[0]}")
PY
