#!/bin/bash
O=gpurun_out/r4_f; mkdir -p $O
python bench.py --train --dtype fp16 --backbone 50 --res 320 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --preheat 0 > $O/c5_nopreheat.json 2> $O/c5_nopreheat.err; echo "rc $?"; tail -2 $O/c5_nopreheat.err
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 python bench.py --train --dtype fp16 --backbone 50 --res 320 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/c5_preheat.json 2> $O/c5_preheat.err; echo "rc $?"; tail -12 $O/c5_preheat.err
python -m pytest tests/test_gpu_h2.py -m gpu -q -s -k "round4" > $O/r4switch.log 2>&1; tail -8 $O/r4switch.log | cut -c1-400
python -m pytest tests/test_gpu_c1w.py tests/test_gpu_c1w_seeds.py -m gpu -q -k "bare or seeds" > $O/parity.log 2>&1; tail -5 $O/parity.log
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $O/kernels.log 2>&1; tail -5 $O/kernels.log
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --steps 20 --no-cpu-baseline --test-cfg FUSE_HEAD_OUT=0 > $O/bench_nofuse.json 2>> $O/bench.err; cut -c1-200 $O/bench_nofuse.json
bash tools/debug/run_timeline.sh r4_f/tl 700 > /dev/null 2>&1; grep -n "dense_glue" -B3 -A17 gpurun_out/r4_f/tl/timeline.txt | tail -22
