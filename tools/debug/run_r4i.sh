#!/bin/bash
O=gpurun_out/r4_i; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_h2.py -m gpu -q -s -k "layer1_kernel" > $O/c64.log 2>&1; tail -8 $O/c64.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_c1w_seeds.py -m gpu -q -s > $O/seeds.log 2>&1; tail -4 $O/seeds.log | cut -c1-400
timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 300 python bench.py --steps 20 --no-cpu-baseline --test-cfg C64_KERNEL=0 > $O/bench_noc64.json 2>> $O/bench.err; cut -c1-200 $O/bench_noc64.json
bash tools/debug/run_timeline.sh r4_i/tl 700 > /dev/null 2>&1; grep -n "stem_pool" -A8 gpurun_out/r4_i/tl/timeline.txt | tail -10; grep -n "dense_glue" -A8 gpurun_out/r4_i/tl/timeline.txt | tail -10
