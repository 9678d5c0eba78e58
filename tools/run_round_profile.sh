#!/bin/bash
# One GPU-box call that produces everything a round's profiles/ entry needs:  bash tools/run_round_profile.sh <tag> [notests]
set -u
TAG=${1:-r2_x}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
if [ "${2:-}" != "notests" ]; then python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log; fi
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --mask-attention mul --no-cpu-baseline --train-leg 0 > $O/bench_mul.json 2>> $O/bench.err
python bench.py --mask-attention mul --cam ycbv --no-cpu-baseline --train-leg 0 > $O/bench_c4_ycbv_mul.json 2>> $O/bench.err
python bench.py --fast x3 --no-cpu-baseline --train-leg 0 > $O/bench_x3.json 2>> $O/bench.err
python bench.py --fast none --no-cpu-baseline --train-leg 0 > $O/bench_fp32mfma.json 2>> $O/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
python bench.py --train --dtype bf16 --steps 30 > $O/bench_train_bf16.json 2>> $O/bench.err
python bench.py --train --dtype bf16 --steps 30 --buckets coarse > $O/bench_train_bf16_coarse_buckets.json 2>> $O/bench.err
python bench.py --train --steps 30 > $O/bench_train_f32.json 2>> $O/bench.err
python bench.py --dtype fp16 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --train --dtype fp16 --steps 30 > $O/bench_train_fp16.json 2>> $O/bench.err
python bench.py --train --dtype fp16 --backbone 50 --res 320 --steps 30 > $O/bench_train_c5_fp16.json 2>> $O/bench.err
python tools/bench_next_rows.py > $O/next_rows.jsonl 2>> $O/bench.err
python bench.py --steps 4000 --trace 200 --no-cpu-baseline --train-leg 0 > $O/bench_sustained.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --train-leg 0 --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python3 tools/conv_stack_fraction.py $O/kernel_stats.csv > $O/conv_stack_fraction.txt; tail -1 $O/conv_stack_fraction.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -- python3 $R/bench.py --train --dtype bf16 --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd $R; f=$(ls $O/proft/*/*kernel_stats.csv | head -1); cp $f $O/train_bf16_kernel_stats.csv
f=$(ls $O/proft/*/*kernel_trace.csv | head -1); python3 tools/train_trace_summary.py $f 1 120 > $O/train_bf16_per_launch.txt; rm -rf $O/proft
bash tools/pmc_bench.sh $TAG --train-leg 0 > $O/pmc.log 2>&1; cp gpurun_out/pmc_$TAG/summary.json $O/pmc_summary.json; cp gpurun_out/pmc_$TAG/summary.md $O/pmc_summary.md; rm -rf gpurun_out/pmc_$TAG/raw_*
for f in bench_mul bench_c4_ycbv_mul bench_x3 bench_fp32mfma bench_bf16 bench_fp16 bench_train_bf16 bench_train_fp16 bench_train_f32 bench_train_c5_fp16 bench_sustained; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d.get('ms_per_step_trace', {}).get('drift_last_vs_first'))"; done
