#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
bash tools/pmc_bench.sh r5_train_bf16 --train --dtype bf16 > $O/pmc.log 2>&1
cp gpurun_out/pmc_r5_train_bf16/summary.json $O/train_bf16_pmc_summary.json; cp gpurun_out/pmc_r5_train_bf16/summary.md $O/train_bf16_pmc_summary.md; rm -rf gpurun_out/pmc_r5_train_bf16/raw_*
head -14 $O/train_bf16_pmc_summary.md
python bench.py --train --dtype bf16 --steps 40 > $O/bench_train_bf16.json 2>/dev/null; cut -c1-300 $O/bench_train_bf16.json
