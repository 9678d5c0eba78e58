#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "row_walking or bit_identical or deterministic or layer" > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 0 1 0 1; do RDPN6D_BN_ROWS=$t python bench.py --train --dtype bf16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bn_rows=$t', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/proft -- python3 $R/bench.py --train --dtype bf16 --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd $R; f=$(ls $O/proft/*/*kernel_trace.csv | head -1); python3 tools/train_trace_summary.py $f 1 400 > $O/train_trace_summary.txt; rm -rf $O/proft
grep -n "bn_apply\|bn_bwd_apply" $O/train_trace_summary.txt
