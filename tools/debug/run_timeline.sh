# timeline (start offset, duration, gap to the previous kernel's end) of the LAST n kernel launches of one bench.py line:
#   bash tools/debug/run_timeline.sh <outdir> <n> <bench.py args...>
R=$(pwd); OUT=$R/gpurun_out/$1; N=$2; shift; shift; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt3
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -- python3 $R/bench.py "$@" --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench.json 2>$OUT/err.log
f=$(ls /tmp/kt3/*/*kernel_trace.csv | head -1)
python3 - "$f" "$N" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  grid {r['Grid_Size_X']:>8}x{r.get('Grid_Size_Y', ''):>4}  {n}")
    prev_end = max(prev_end, e)
PY
