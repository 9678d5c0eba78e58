for L in 1 0; do for B in 1 2 4; do
RDPN6D_H2_SPLIT_LONGK=$L python bench.py --no-cpu-baseline --batch $B --steps 300 --preheat 0.5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('longk=$L B=$B', d['ms_per_step'])"
done; done
python -m pytest tests/test_gpu_c1w.py tests/test_gpu_h2.py -m gpu -x -q -k "bare_tolerance or fp32_accuracy" 2>&1 | tail -2
