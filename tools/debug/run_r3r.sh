#!/bin/bash
O=gpurun_out/r3_r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_h2.py tests/test_gpu_c1w.py -x -q -k "fp32_accuracy or bare_tolerance or cancellation" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer | tee $O/conv.log
RDPN6D_H2_PP=0 timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer | tee -a $O/conv.log
for i in 1 2; do timeout 600 python bench.py --steps 150 --no-cpu-baseline 2>/dev/null | cut -c1-150; done | tee $O/bench.log
