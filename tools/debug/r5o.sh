#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2>/dev/null
cd $R; f=$(ls $O/prof/*/*kernel_trace.csv | head -1); python3 tools/train_trace_summary.py $f 1 400 > $O/infer_trace_summary.txt; rm -rf $O/prof
head -3 $O/infer_trace_summary.txt; cut -c1-150 $O/bench.json
