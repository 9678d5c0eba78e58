#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r5_fin; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json
python bench.py --train --dtype bf16 --steps 40 > $O/bench_train_bf16.json 2>>$O/bench.err; cut -c1-220 $O/bench_train_bf16.json
python bench.py --train --dtype fp16 --steps 40 > $O/bench_train_fp16.json 2>>$O/bench.err; cut -c1-220 $O/bench_train_fp16.json
python bench.py --train --dtype fp16 --backbone 50 --res 320 --steps 30 > $O/bench_train_c5_fp16.json 2>>$O/bench.err; cut -c1-220 $O/bench_train_c5_fp16.json
