#!/bin/bash
O=gpurun_out/r3_h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q -k "fp32_accuracy" > $O/h2_tests.log 2>&1; tail -15 $O/h2_tests.log
for cfg in "0 1 4" "1 0 4" "1 1 4" "1 2 4" "1 1 3"; do set -- $cfg; echo "PP=$1 PM=$2 NST=$3"; RDPN6D_H2_PP=$1 RDPN6D_H2_PP_PM=$2 RDPN6D_H2_PP_NST=$3 timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer; done | tee $O/conv.log
echo "PP shape 0 forced (128x128 for layer2)"; RDPN6D_H2_PP_SHAPE=0 timeout 300 python tools/bench_conv_h2.py 2>&1 | grep layer2 | tee -a $O/conv.log
for pp in 0 1; do RDPN6D_H2_PP=$pp timeout 600 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | cut -c1-160; done | tee $O/bench.log
timeout 900 python -m pytest tests/test_gpu_c1w.py -x -q -k "bare_tolerance" > $O/c1w.log 2>&1; tail -4 $O/c1w.log
