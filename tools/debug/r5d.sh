mkdir -p gpurun_out/r5_d; O=gpurun_out/r5_d
python -m pytest tests/test_gpu_h2.py -m gpu -x -q -s -k "fp32_accuracy" > $O/h2_tests.log 2>&1; tail -3 $O/h2_tests.log; grep -c "bit-identical" $O/h2_tests.log
python bench.py --steps 50 --no-cpu-baseline > $O/bench_bfg.json 2> $O/bench.err
RDPN6D_H2_BFG=0 python bench.py --steps 50 --no-cpu-baseline > $O/bench_nobfg.json 2>> $O/bench.err
RDPN6D_H2_BFG=2 python bench.py --steps 50 --no-cpu-baseline > $O/bench_bfg2.json 2>> $O/bench.err
RDPN6D_H2_BFG_PM=0 python bench.py --steps 50 --no-cpu-baseline > $O/bench_bfg_pm0.json 2>> $O/bench.err
for f in bench_bfg bench_nobfg bench_bfg2 bench_bfg_pm0; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python3 tools/conv_stack_fraction.py $O/kernel_stats.csv > $O/conv_stack_fraction.txt; cat $O/conv_stack_fraction.txt
cd /tmp
RDPN6D_H2_BFG=2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof2 -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof2/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats_bfg2.csv; rm -rf $O/prof2
python3 tools/conv_stack_fraction.py $O/kernel_stats_bfg2.csv > $O/conv_stack_fraction_bfg2.txt; cat $O/conv_stack_fraction_bfg2.txt
RDPN6D_PROBE=1 python rdpn6d_amd/build.py --force > $O/probe_build.log 2>&1; tail -1 $O/probe_build.log
python tools/probe_stem.py > $O/probe_stem.log 2>&1; tail -2 $O/probe_stem.log
