#!/bin/bash
# round-4 first box call: the new seed-independent parity test, the DDP-wrapper test, the touched host-semantics tests, one bench line
O=gpurun_out/r4_a; mkdir -p $O
python -m pytest tests/test_gpu_c1w_seeds.py -m gpu -x -q -s > $O/seeds.log 2>&1; tail -25 $O/seeds.log
python -m pytest tests/test_gpu_host_semantics.py -m gpu -q > $O/host.log 2>&1; tail -15 $O/host.log
python -m pytest tests/test_gpu_c1w.py tests/test_gpu_train.py -m gpu -q -x > $O/train.log 2>&1; tail -5 $O/train.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
