#!/bin/bash
O=gpurun_out/r3_v; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
for sk in 0 6000 12000 18000 24000; do echo "SKEW=$sk"; RDPN6D_H2_PP=0 RDPN6D_H2_SKEW=$sk timeout 300 python tools/bench_conv_h2.py 2>&1 | grep "layer"; done | tee $O/skew.log
