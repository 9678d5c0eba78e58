set -u
R=$(pwd); O=$R/gpurun_out/r2_g; mkdir -p $O
python -m pytest tests/test_gpu_h2.py -q -s -k "stem_pool" 2>&1 | grep -v amdgpu | tail -8
python -m pytest tests/test_gpu_c1w.py tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -x -k "c1w or fast_forms or b64 or golden or other_resnet or c4 or generalised" > $O/tests.log 2>&1; tail -4 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench_h2.json 2> $O/bench.err; cut -c1-250 $O/bench_h2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd $R; f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-150; rm -rf $O/prof
