#!/bin/bash
O=gpurun_out/r3_t; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
for sg in 0 500 1000 2000 4000; do echo "STAGGER=$sg"; RDPN6D_H2_STAGGER=$sg timeout 300 python tools/bench_conv_h2.py 2>&1 | grep "layer1\|head"; done | tee $O/stagger.log
