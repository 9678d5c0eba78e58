R=$(pwd); cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  rm -rf /tmp/pp; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp -- python3 $R/tools/debug/stem_probe.py > /dev/null 2>&1
  f=$(ls /tmp/pp/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "stem_pool" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(k, sum(v) / len(v))
PY
done
