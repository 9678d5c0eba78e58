#!/bin/bash
# round 3, first box: GPU suite + smoke + headline bench (sync / deferred range check) + trunk layer microbench
O=gpurun_out/r3_a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 100 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
python bench.py --steps 100 --range-check deferred --no-cpu-baseline > $O/bench_deferred.json 2>> $O/bench.err; cut -c1-200 $O/bench_deferred.json
python bench.py --steps 100 --no-cpu-baseline > $O/bench2.json 2>> $O/bench.err; cut -c1-200 $O/bench2.json
python bench.py --steps 200 --batch 1 --no-cpu-baseline > $O/bench_b1.json 2>> $O/bench.err; cut -c1-200 $O/bench_b1.json
python bench.py --steps 200 --batch 1 --range-check deferred --no-cpu-baseline > $O/bench_b1_deferred.json 2>> $O/bench.err; cut -c1-200 $O/bench_b1_deferred.json
python tools/bench_conv_h2.py > $O/conv_h2.log 2>&1; cat $O/conv_h2.log
