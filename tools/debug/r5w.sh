#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_w; mkdir -p $O
for t in 0 1 2 3 0 1 2 3; do RDPN6D_BN_NT=$t python bench.py --train --dtype bf16 --steps 40 --no-cpu-baseline 2>>$O/bench.err | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bn_nt=$t', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
