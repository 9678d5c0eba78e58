cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in "128,128" "256,256"; do
 i=0
 for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" ; do
  i=$((i+1))
  TILE=$T REPS=12 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/$T/p$i -- python3 $R/tools/steady_conv_bf16.py > /dev/null 2>&1
 done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for T in ("128,128", "256,256"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcb/{T}/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "conv_igemm_bf16" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("TILE", T)
    for k in sorted(agg):
        v = agg[k][2:]  # skip warm launches
        print(f"  {k:42s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
