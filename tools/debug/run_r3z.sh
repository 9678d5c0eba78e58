#!/bin/bash
O=gpurun_out/r3_z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_c1w.py -x -q -k "stem or bare_tolerance" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do timeout 600 python bench.py --steps 150 --no-cpu-baseline 2>/dev/null | cut -c1-150; done | tee $O/bench.log
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
python tools/probe_stem.py 2>&1 | grep -v amdgpu | tee $O/probe.log
