mkdir -p gpurun_out/r5_f; O=gpurun_out/r5_f; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$B -- python3 $R/bench.py --no-cpu-baseline --batch $B --steps 200 --preheat 0.5 > /dev/null 2>&1
f=$(ls $R/$O/prof$B/*/*kernel_stats.csv | head -1); cp $f $R/$O/kernel_stats_b$B.csv; rm -rf $R/$O/prof$B
done
cd $R; python bench.py --no-cpu-baseline --batch 1 --steps 300 --preheat 0.5 2>/dev/null | cut -c1-200
