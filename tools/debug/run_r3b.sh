#!/bin/bash
O=gpurun_out/r3_b; mkdir -p $O
python -m pytest tests/test_gpu_host_semantics.py tests/test_gpu_pnp.py -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
for s in 0 1 2; do RDPN6D_H2_SCHED=$s python tools/bench_conv_h2.py > $O/conv_h2_s$s.log 2>&1; echo "SCHED=$s"; grep -v amdgpu $O/conv_h2_s$s.log; done
for s in 0 1; do RDPN6D_H2_SCHED=$s RDPN6D_H2_NST=3 python tools/bench_conv_h2.py > $O/conv_h2_s${s}_n3.log 2>&1; echo "SCHED=$s NST=3"; grep -v amdgpu $O/conv_h2_s${s}_n3.log; done
for s in 0 1; do RDPN6D_H2_SCHED=$s python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | cut -c1-160; done
