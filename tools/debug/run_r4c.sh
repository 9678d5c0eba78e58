#!/bin/bash
O=gpurun_out/r4_c; mkdir -p $O
python tools/debug/amp_stage_diff.py fp16 4 > $O/amp_stage_fp16.log 2>&1; head -60 $O/amp_stage_fp16.log
python tools/debug/amp_stage_diff.py bf16 4 > $O/amp_stage_bf16.log 2>&1
python -m pytest tests/test_gpu_c1w_seeds.py -m gpu -q -s > $O/seeds.log 2>&1; tail -4 $O/seeds.log
python -m pytest tests/test_gpu_c1w.py tests/test_gpu_kernels.py -m gpu -q -x -k "bare_tolerance or teacher_forced or fast_forms" > $O/pnp_h2.log 2>&1; tail -15 $O/pnp_h2.log
python tools/debug/ransac_phases.py > $O/ransac_phases.log 2>&1; cat $O/ransac_phases.log
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --steps 20 --no-cpu-baseline --test-cfg PNP_H2=0 > $O/bench_nopnph2.json 2>> $O/bench.err; cut -c1-200 $O/bench_nopnph2.json
bash tools/debug/run_timeline.sh r4_c/tl 40 > /dev/null 2>&1; cat gpurun_out/r4_c/tl/timeline.txt
python -m pytest tests/test_gpu_fp16.py -m gpu -q -s -k "gradscaler" > $O/gs.log 2>&1; tail -8 $O/gs.log
