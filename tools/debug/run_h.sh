R=$(pwd); cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the last full step: from last stem_pool to the end
idx=[i for i,r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n=r["Kernel_Name"]
    short=n.split("(")[0].replace("void ","").replace("(anonymous namespace)::","")[:60]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  {short}  grid {r.get("Grid_Size_X","?")}')
print("step span", (int(rows[b]["Start_Timestamp"])-t0)/1e3, "us")
PY
