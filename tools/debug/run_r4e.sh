#!/bin/bash
O=gpurun_out/r4_e; mkdir -p $O
python -m pytest tests/test_gpu_workloads.py tests/test_gpu_pnp.py -m gpu -q > $O/workloads.log 2>&1; tail -6 $O/workloads.log
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "ransac or kabsch or pnp or teacher" > $O/kernels.log 2>&1; tail -4 $O/kernels.log
python -m pytest tests/test_gpu_c1w.py -m gpu -q -s -k "every_layer" > $O/c1w.log 2>&1; tail -5 $O/c1w.log | cut -c1-900
python tools/debug/ransac_phases.py 2>&1 | grep -v amdgpu | tee $O/ransac_phases.log
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
RDPN6D_RANSAC_NO_SPLIT=1 python bench.py --steps 20 --no-cpu-baseline > $O/bench_nosplit.json 2>> $O/bench.err; cut -c1-200 $O/bench_nosplit.json
bash tools/debug/run_timeline.sh r4_e/tl 700 > /dev/null 2>&1; grep -n "dense_glue" -B2 -A26 gpurun_out/r4_e/tl/timeline.txt | tail -32
