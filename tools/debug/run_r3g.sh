#!/bin/bash
O=gpurun_out/r3_g; mkdir -p $O
for t in "64,128" "128,64" "64,64"; do for n in 2 3; do echo "SCHED=0 NST=$n TILE=$t"; RDPN6D_H2_SCHED=0 RDPN6D_H2_NST=$n RDPN6D_H2_TILE=$t python tools/bench_conv_h2.py 2>&1 | grep layer; done; done | tee $O/conv_tiles.log
for t in "64,128" "128,64"; do echo "SCHED=1 NST=2 TILE=$t"; RDPN6D_H2_SCHED=1 RDPN6D_H2_NST=2 RDPN6D_H2_TILE=$t python tools/bench_conv_h2.py 2>&1 | grep layer; done | tee -a $O/conv_tiles.log
