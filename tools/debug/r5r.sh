#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_r; mkdir -p $O
timeout 900 python -m pytest tests/test_ranger.py tests/test_gpu_host_semantics.py tests/test_gpu_fp16.py -q -x -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in bf16 fp16 fp16; do python bench.py --train --dtype $t --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'], d.get('steps_skipped_for_overflow'), d.get('loss_scale_final'))"; done | tee $O/ab.txt
