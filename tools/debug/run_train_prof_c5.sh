# kernel stats of the C5-shape training step (ResNet-50, 320x320, fp16): bash tools/debug/run_train_prof_c5.sh <outdir-under-gpurun_out>
R=$(pwd); OUT=$R/gpurun_out/${1:-train_prof_c5}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python3 $R/bench.py --train --dtype fp16 --backbone 50 --res 320 --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench.json 2>$OUT/err.log
f=$(ls /tmp/kt5/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
steps=next(int(r["Calls"]) for r in rows if r["Name"].startswith("ranger_update_kernel"))
print(f"{steps} steps profiled; total kernel ms per step", tot/1e6/steps)
for r in rows[:40]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:74]
    print(f'{int(r["TotalDurationNs"])/1e6/steps:7.3f} ms/step {r["Percentage"]:>6}%  calls/step {int(r["Calls"])/steps:6.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  {n}')
PY
