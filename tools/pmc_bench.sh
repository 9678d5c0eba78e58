#!/bin/bash
# rocprofv3 --pmc passes over bench.py (one counter set per pass, kernel trace only - the combination gpurun allows),
# then tools/summarize_pmc.py.   usage: tools/pmc_bench.sh <tag> [bench.py args...]      (run on the GPU box)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  name=${set%% *}
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/raw_$name -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --preheat 0 "$@" > /dev/null 2>&1
  f=$(ls $OUT/raw_$name/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_counter_collection.csv
done
cd $R
python3 tools/summarize_pmc.py $OUT $OUT/summary
