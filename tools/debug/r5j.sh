mkdir -p gpurun_out/r5_j; O=gpurun_out/r5_j
python -m pytest tests/test_gpu_mask_types.py tests/test_gpu_train.py tests/test_gpu_c1w.py -m gpu -x -q -s > $O/tests.log 2>&1; tail -4 $O/tests.log; grep "train BCE\|train CE" $O/tests.log | grep "HIP vs" 
python bench.py --train --dtype bf16 --steps 30 2>/dev/null | cut -c1-160
