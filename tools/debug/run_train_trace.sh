# ordered kernel list of one training step: bash tools/debug/run_train_trace.sh <dtype> <outname>
R=$(pwd); DT=${1:-bf16}; OUT=$R/gpurun_out/${2:-train_trace}.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --train --dtype $DT --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1)
python3 - "$f" > $OUT <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "repack_kernel" in r["Kernel_Name"]]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]["Start_Timestamp"])
prev_end=t0
for r in rows[a:b]:
    n=r["Kernel_Name"]
    short=n.replace("void ","").replace("(anonymous namespace)::","")[:70]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:9.1f} us gap {(s-prev_end)/1e3:6.1f} +{(e-s)/1e3:8.1f} us  {short}  grid {r.get("Grid_Size_X","?")},{r.get("Grid_Size_Y","?")}')
    prev_end=e
print("step span", (int(rows[b]["Start_Timestamp"])-t0)/1e3, "us", "kernels", b-a)
PY
tail -1 $OUT
