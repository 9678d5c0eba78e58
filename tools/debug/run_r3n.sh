#!/bin/bash
O=gpurun_out/r3_n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_semantics.py tests/test_gpu_c1w.py -x -q -k "h2_range or bare_tolerance" > $O/tests.log 2>&1; tail -3 $O/tests.log
for rc in sync deferred sync deferred; do timeout 600 python bench.py --steps 150 --no-cpu-baseline --range-check $rc 2>/dev/null | cut -c1-150; done | tee $O/bench.log
for rc in sync deferred; do timeout 600 python bench.py --steps 300 --batch 1 --no-cpu-baseline --range-check $rc 2>/dev/null | cut -c1-150; done | tee -a $O/bench.log
for rc in sync deferred; do timeout 600 python bench.py --steps 300 --batch 8 --no-cpu-baseline --range-check $rc 2>/dev/null | cut -c1-150; done | tee -a $O/bench.log
