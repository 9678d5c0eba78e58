#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -q -x -k "wgrad" > $O/tests.log 2>&1; tail -3 $O/tests.log
for t in 0 1; do echo "SQUARE=$t"; RDPN6D_WGRAD_SQUARE=$t python tools/bench_wgrad_bf16.py 2>&1 | grep -v "^check" | head -3; done | tee $O/micro.txt
for t in 0 1 0 1; do RDPN6D_WGRAD_SQUARE=$t python bench.py --train --dtype bf16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('square=$t', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
