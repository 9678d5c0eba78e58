mkdir -p gpurun_out/r5_c; O=gpurun_out/r5_c
python -m pytest tests/test_gpu_h2.py -m gpu -x -q -k "stem" > $O/stem_tests.log 2>&1; tail -3 $O/stem_tests.log
python bench.py --steps 50 --no-cpu-baseline > $O/bench_v2.json 2> $O/bench.err
RDPN6D_STEM_V1=1 python bench.py --steps 50 --no-cpu-baseline > $O/bench_v1.json 2>> $O/bench.err
python bench.py --steps 50 --no-cpu-baseline > $O/bench_v2b.json 2>> $O/bench.err
for f in bench_v2 bench_v1 bench_v2b; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python3 tools/conv_stack_fraction.py $O/kernel_stats.csv > $O/conv_stack_fraction.txt; cat $O/conv_stack_fraction.txt
python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
