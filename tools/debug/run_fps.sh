#!/bin/bash
O=gpurun_out/${1:-fps}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fps" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python tools/bench_next_rows.py > $O/next_rows.jsonl 2> $O/err.log; cut -c1-700 $O/next_rows.jsonl; tail -3 $O/err.log
