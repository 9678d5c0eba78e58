# kernel stats of one bench.py line:  bash tools/debug/run_kstats.sh <outdir-under-gpurun_out> <bench.py args...>
R=$(pwd); OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py "$@" --no-cpu-baseline --steps 20 --warmup 4 > $OUT/bench.json 2>$OUT/err.log
f=$(ls /tmp/kt/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:32]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:80]
    print(f'{int(r["TotalDurationNs"])/1e6:8.3f} ms {r["Percentage"]:>6}%  calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  {n}')
PY
tail -c 300 $OUT/bench.json
