#!/bin/bash
O=gpurun_out/r3_j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_host_semantics.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for t in "128,64" "256,64"; do echo "TILE=$t"; RDPN6D_H2_TILE=$t timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer1; done | tee $O/conv.log
echo default; timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer | tee -a $O/conv.log
for t in "128,64" "256,64" "128,64" "256,64"; do RDPN6D_H2_TILE=$t timeout 600 python bench.py --steps 150 --no-cpu-baseline 2>/dev/null | cut -c1-160; done | tee $O/bench.log
