# kernel stats of the AMP training step: bash tools/debug/run_train_prof.sh <dtype> <outdir-under-gpurun_out>
R=$(pwd); DT=${1:-bf16}; OUT=$R/gpurun_out/${2:-train_prof}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --train $( [ "$DT" = f32 ] || echo --dtype $DT ) --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench.json 2>$OUT/err.log
f=$(ls /tmp/kt/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
steps=next(int(r["Calls"]) for r in rows if r["Name"].startswith("ranger_update_kernel"))  # one optimizer launch per step (warm-up, pre-heat and timed steps alike)
print(f"{steps} steps profiled; total kernel ms per step", tot/1e6/steps)
for r in rows[:45]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]
    print(f'{int(r["TotalDurationNs"])/1e6/steps:7.3f} ms/step {r["Percentage"]:>6}%  calls/step {int(r["Calls"])/steps:6.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  {n}')
PY
tail -c 400 $OUT/bench.json
