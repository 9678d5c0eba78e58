#!/bin/bash
O=gpurun_out/r3_d; mkdir -p $O
python -m pytest tests/test_gpu_host_semantics.py -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
for cfg in "0 0" "0 3" "3 3" "1 3"; do set -- $cfg; echo "SCHED=$1 NST=$2"; RDPN6D_H2_SCHED=$1 RDPN6D_H2_NST=$2 python tools/bench_conv_h2.py 2>&1 | grep layer; done | tee $O/conv.log
for t in "64,64" "128,64" "64,128" "128,128"; do echo "SCHED=3 NST=3 TILE=$t"; RDPN6D_H2_SCHED=3 RDPN6D_H2_NST=3 RDPN6D_H2_TILE=$t python tools/bench_conv_h2.py 2>&1 | grep layer; done | tee $O/conv_tiles.log
for cfg in "0 0" "3 3"; do set -- $cfg; RDPN6D_H2_SCHED=$1 RDPN6D_H2_NST=$2 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | cut -c1-160; done | tee $O/bench.log
