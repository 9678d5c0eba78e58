#!/bin/bash
O=gpurun_out/r3_l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_pipeline.py -x -q -m gpu -k "hip_graph or b64_properties or frames_to_pose" > $O/tests.log 2>&1; tail -4 $O/tests.log
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
echo "head layer, normal"; timeout 300 python tools/bench_conv_h2.py 2>&1 | grep head | tee $O/abl.log
echo "head layer, A half-tiles of taps 1..8 out of range (no L2 traffic)"; RDPN6D_H2_ABL_A=1 timeout 300 python tools/bench_conv_h2.py 2>&1 | grep head | tee -a $O/abl.log
