python -m pytest tests/test_gpu_h2.py tests/test_gpu_c1w.py -m gpu -x -q -k "fp32_accuracy or bare_tolerance" 2>&1 | tail -3
for B in 1 2 4 8 15; do
python bench.py --no-cpu-baseline --batch $B --steps 300 --preheat 0.5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B', d['ms_per_step'], d['value'])"
done
