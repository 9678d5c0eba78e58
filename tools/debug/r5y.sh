#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_y; mkdir -p $O
timeout 900 python -m pytest tests/test_ranger.py tests/test_gpu_host_semantics.py -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 1 2 3; do python bench.py --train --dtype bf16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
python bench.py --train --dtype fp16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp16', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
