#!/bin/bash
O=gpurun_out/probe; mkdir -p $O
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
for pm in 2; do echo "PM=$pm"; RDPN6D_H2_PP_PM=$pm python tools/probe_h2_pp.py 2>&1 | grep -v amdgpu; done | tee $O/probe.log
