#!/bin/bash
# stage-count experiments inside the real steps, log under gpurun_out/<tag>/
O=gpurun_out/${1:-nst}; mkdir -p $O
for k in 24 1000; do
  RDPN6D_CONV_LP_NST3_K=$k python bench.py --train --dtype bf16 --steps 30 --no-cpu-baseline 2>>$O/err.log | cut -c1-230 | sed "s/^/train bf16 NST3_K=$k: /"
done
RDPN6D_CONV_LP_NST3_K=24 python bench.py --dtype bf16 --no-cpu-baseline 2>>$O/err.log | cut -c1-200 | sed "s/^/infer bf16 NST3_K=24: /"
RDPN6D_CONV_LP_NST3_K=1000 python bench.py --dtype bf16 --no-cpu-baseline 2>>$O/err.log | cut -c1-200 | sed "s/^/infer bf16 NST3_K=1000: /"
python bench.py --no-cpu-baseline 2>>$O/err.log | cut -c1-200 | sed "s/^/headline default: /"
RDPN6D_H2_NST=3 python bench.py --no-cpu-baseline 2>>$O/err.log | cut -c1-200 | sed "s/^/headline H2_NST=3: /"
RDPN6D_H2_NST=2 python bench.py --no-cpu-baseline 2>>$O/err.log | cut -c1-200 | sed "s/^/headline H2_NST=2: /"
