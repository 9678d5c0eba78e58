#!/bin/bash
O=gpurun_out/r4_b; mkdir -p $O
python -m pytest tests/test_gpu_c1w_seeds.py -m gpu -q -s > $O/seeds.log 2>&1; tail -4 $O/seeds.log
python -m pytest tests/test_gpu_c1w.py -m gpu -q -s -k "amp_step" > $O/amp_forced.log 2>&1; tail -12 $O/amp_forced.log
python -m pytest tests/test_gpu_fp16.py -m gpu -q -s -k "c5_at" > $O/c5.log 2>&1; tail -5 $O/c5.log
python -m pytest tests/test_gpu_host_semantics.py -m gpu -q -s -k "vis or ddp or pnp" > $O/host.log 2>&1; tail -8 $O/host.log
python tools/debug/gradscaler_probe.py 2e-3 8 > $O/gs_probe_fused.log 2>&1
RDPN6D_BN_FUSE_STATS=0 RDPN6D_BN_FUSE_BWD=0 RDPN6D_MFMA_STEM=0 python tools/debug/gradscaler_probe.py 2e-3 8 > $O/gs_probe_unfused.log 2>&1
tail -30 $O/gs_probe_fused.log
python bench.py --train --dtype bf16 --steps 20 > $O/bench_train_bf16.json 2> $O/bench.err; cat $O/bench_train_bf16.json | cut -c1-1500
python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2>> $O/bench.err; cut -c1-400 $O/bench.json
