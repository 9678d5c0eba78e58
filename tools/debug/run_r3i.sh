#!/bin/bash
O=gpurun_out/r3_i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_host_semantics.py tests/test_gpu_pnp.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for pp in 0 1; do echo "PP=$pp"; RDPN6D_H2_PP=$pp timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer; done | tee $O/conv.log
for pp in 0 1 0 1; do RDPN6D_H2_PP=$pp timeout 600 python bench.py --steps 150 --no-cpu-baseline 2>/dev/null | cut -c1-160; done | tee $O/bench.log
