# ordered kernel list of one inference step at a small batch: bash tools/debug/run_small_trace.sh <B>
R=$(pwd); B=${1:-4}; OUT=$R/gpurun_out/small_trace_B$B.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --batch $B --steps 6 --warmup 3 > /dev/null 2>&1
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1)
python3 - "$f" > $OUT <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "stem" in r["Kernel_Name"]]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]["Start_Timestamp"]); prev=t0; busy=0
for r in rows[a:b]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"]); busy+=e-s
    n=r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","")[:60]
    print(f'{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:5.1f} +{(e-s)/1e3:7.1f} {n} grid {r.get("Grid_Size_X","?")},{r.get("Grid_Size_Y","?")}')
    prev=e
print("step span", (int(rows[b]["Start_Timestamp"])-t0)/1e3, "us; kernels", b-a, "; busy", busy/1e3)
PY
tail -1 $OUT
