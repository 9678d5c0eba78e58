#!/bin/bash
O=gpurun_out/r4_k; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build_probe.log 2>&1; tail -1 $O/build_probe.log
timeout 300 python tools/debug/probe_c64.py 2>&1 | grep -v amdgpu | tee $O/probe_c64.log | head -40
