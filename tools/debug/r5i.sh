mkdir -p gpurun_out/r5_i; O=gpurun_out/r5_i
python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "side_stream or deterministic" > $O/tests.log 2>&1; tail -4 $O/tests.log
for side in 1 0 1 0; do
RDPN6D_WGRAD_SIDE=$side python bench.py --train --dtype bf16 --steps 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side=$side bf16', d['ms_per_step'], d['value'])"
done
for side in 1 0; do
RDPN6D_WGRAD_SIDE=$side python bench.py --train --dtype fp16 --steps 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side=$side fp16', d['ms_per_step'], d['value'])"
done
