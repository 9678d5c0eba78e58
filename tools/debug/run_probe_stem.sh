#!/bin/bash
O=gpurun_out/probe; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
python tools/probe_stem.py 2>&1 | grep -v amdgpu | tee $O/probe.log
