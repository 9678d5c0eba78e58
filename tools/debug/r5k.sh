mkdir -p gpurun_out/r5_k; O=gpurun_out/r5_k
python -m pytest tests/test_gpu_h2.py -m gpu -x -q -k "fused_global_max or folded_global_max or round4_plan" > $O/tests.log 2>&1; tail -4 $O/tests.log
python bench.py --steps 50 --no-cpu-baseline > $O/bench_on.json 2> $O/bench.err
python bench.py --steps 50 --no-cpu-baseline --test-cfg FUSE_GLOBAL_MAX=0 > $O/bench_off.json 2>> $O/bench.err
python bench.py --steps 50 --no-cpu-baseline > $O/bench_on2.json 2>> $O/bench.err
python bench.py --steps 50 --no-cpu-baseline --test-cfg FUSE_GLOBAL_MAX=0 > $O/bench_off2.json 2>> $O/bench.err
for f in bench_on bench_off bench_on2 bench_off2; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches_per_step'], d['roofline']['algorithmic_gflop_per_launch'])"; done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python3 tools/conv_stack_fraction.py $O/kernel_stats.csv > $O/conv_stack_fraction.txt; cat $O/conv_stack_fraction.txt
grep -i "colmax_decode\|global_max" $O/kernel_stats.csv | cut -c1-160
