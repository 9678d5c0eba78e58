#!/bin/bash
O=gpurun_out/r3_c; mkdir -p $O
python -m pytest tests/test_gpu_host_semantics.py tests/test_gpu_pnp.py -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
RDPN6D_H2_SCHED=9 RDPN6D_H2_NST=3 python tools/probe_h2_tile.py 2>&1 | grep -v amdgpu | tee $O/probe.log
RDPN6D_H2_SCHED=9 RDPN6D_H2_NST=3 RDPN6D_H2_TILE=64,64 python tools/probe_h2_tile.py 2>&1 | grep -v amdgpu | tee $O/probe_64x64.log
RDPN6D_H2_SCHED=9 RDPN6D_H2_NST=3 RDPN6D_H2_TILE=128,64 python tools/probe_h2_tile.py 2>&1 | grep -v amdgpu | tee $O/probe_128x64.log
