#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_fp16.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 1 2; do python bench.py --train --dtype bf16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
python bench.py --train --steps 30 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/proft -- python3 $R/bench.py --train --dtype bf16 --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd $R; f=$(ls $O/proft/*/*kernel_trace.csv | head -1); python3 tools/train_trace_summary.py $f 1 400 > $O/train_trace_summary.txt; rm -rf $O/proft
grep -n "stem\|wgrad_bf16_kernel<64, 64" $O/train_trace_summary.txt
