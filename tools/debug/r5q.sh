#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_q; mkdir -p $O
for t in bf16 fp16; do python bench.py --train --dtype $t --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for t in bf16 fp16; do
rocprofv3 --kernel-trace --output-format csv -d $O/proft -- python3 $R/bench.py --train --dtype $t --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
f=$(ls $O/proft/*/*kernel_trace.csv | head -1); python3 $R/tools/train_trace_summary.py $f 1 400 > $O/train_trace_$t.txt; rm -rf $O/proft
done
