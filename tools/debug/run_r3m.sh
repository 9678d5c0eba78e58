#!/bin/bash
O=gpurun_out/r3_m; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
for m in 0 1 2 0 2; do echo "ABL_A=$m"; RDPN6D_H2_ABL_A=$m timeout 300 python tools/bench_conv_h2.py 2>&1 | grep head; done | tee $O/abl.log
