mkdir -p gpurun_out/r5_b
python -m pytest tests -m gpu -x -q > gpurun_out/r5_b/gpu_tests.log 2>&1; tail -3 gpurun_out/r5_b/gpu_tests.log
for B in 1 4 8 15; do
  python bench.py --batch $B --steps 200 --warmup 5 --preheat 1 --no-cpu-baseline > gpurun_out/r5_b/lat_eager_$B.json 2>> gpurun_out/r5_b/lat.err
  python bench.py --batch $B --steps 200 --warmup 5 --preheat 1 --no-cpu-baseline --graph > gpurun_out/r5_b/lat_graph_$B.json 2>> gpurun_out/r5_b/lat.err
done
for f in gpurun_out/r5_b/lat_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'])"; done
