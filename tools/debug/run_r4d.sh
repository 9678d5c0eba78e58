#!/bin/bash
O=gpurun_out/r4_d; mkdir -p $O
python -m pytest tests/test_gpu_c1w.py -m gpu -q -s -k "bare_tolerance or every_layer" > $O/c1w.log 2>&1; tail -12 $O/c1w.log | cut -c1-1500
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "teacher_forced or fast_forms or ransac or pnp" > $O/kernels.log 2>&1; tail -6 $O/kernels.log
python -m pytest tests/test_gpu_host_semantics.py tests/test_gpu_pnp.py tests/test_gpu_workloads.py -m gpu -q > $O/host.log 2>&1; tail -6 $O/host.log
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --steps 20 --no-cpu-baseline --test-cfg PNP_H2=0 > $O/bench_nopnph2.json 2>> $O/bench.err; cut -c1-200 $O/bench_nopnph2.json
bash tools/debug/run_timeline.sh r4_d/tl 130 > /dev/null 2>&1; grep -n "dense_glue" -B3 -A28 gpurun_out/r4_d/tl/timeline.txt | tail -45
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build_probe.log 2>&1; tail -1 $O/build_probe.log
python tools/debug/ransac_phases.py 2>&1 | grep -v amdgpu | tee $O/ransac_phases.log
