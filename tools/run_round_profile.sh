set -u
R=$(pwd); O=$R/gpurun_out/r2_a; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-300
python bench.py --mask-attention mul --no-cpu-baseline > $O/bench_mul.json 2>> $O/bench.err
python bench.py --train --dtype bf16 --steps 30 > $O/bench_train_bf16.json 2>> $O/bench.err
python bench.py --train --steps 30 > $O/bench_train_f32.json 2>> $O/bench.err
python bench.py --steps 2000 --trace 100 --no-cpu-baseline > $O/bench_sustained.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; head -25 $O/kernel_stats.csv | cut -c1-160
bash tools/pmc_bench.sh r2_a > $O/pmc.log 2>&1; cp gpurun_out/pmc_r2_a/summary.* $O/ 2>/dev/null; rm -rf $O/prof gpurun_out/pmc_r2_a/raw_*
