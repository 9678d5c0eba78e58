#!/bin/bash
# grouped weight gradient: tests + training bench lines, log under gpurun_out/<tag>/
O=gpurun_out/${1:-wg}; mkdir -p $O
python -m pytest tests/test_gpu_train.py -m gpu -q -k "grouped or deterministic or wgrad" > $O/tests.log 2>&1; tail -15 $O/tests.log
python bench.py --train --dtype bf16 --steps 30 --no-cpu-baseline > $O/bench_train_bf16.json 2>$O/err.log; cut -c1-260 $O/bench_train_bf16.json
RDPN6D_GROUP_WGRAD=0 python bench.py --train --dtype bf16 --steps 30 --no-cpu-baseline > $O/bench_train_bf16_ungrouped.json 2>>$O/err.log; cut -c1-260 $O/bench_train_bf16_ungrouped.json
