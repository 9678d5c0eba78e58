#!/bin/bash
O=gpurun_out/r3_f; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/probe/clock_probe.hip 2>/dev/null && /tmp/clock_probe | tee $O/clock.log
for b in 8 16 32 64 128; do echo "B=$b"; B=$b python tools/bench_conv_h2.py 2>&1 | grep "layer3\|layer1"; done | tee $O/bsweep.log
