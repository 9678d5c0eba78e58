#!/bin/bash
O=gpurun_out/r4_h; mkdir -p $O
for v in "" "FUSE_HEAD_OUT=0" "PNP_H2=0" "C64_KERNEL=0" "FUSE_HEAD_OUT=0,PNP_H2=0,C64_KERNEL=0"; do
  RDPN6D_SEEDS_TEST_CFG="$v" timeout 300 python -m pytest tests/test_gpu_c1w_seeds.py -m gpu -q -s -k "h2-mul or h2-none" > $O/seeds_$(echo $v | tr ',=' '__').log 2>&1
  echo "== $v"; grep -o "\[seeds h2 [a-z]*\].*pose worst[^;]*; slots[^\n]*" $O/seeds_$(echo $v | tr ',=' '__').log | sed 's/maps max-abs over 64 slots: //' | cut -c1-330
done
timeout 900 python tools/debug/c5_nan_probe.py 400 full > $O/c5_nan.log 2>&1; tail -12 $O/c5_nan.log | cut -c1-200
