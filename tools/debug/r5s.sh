#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/r5_s; mkdir -p $O
for b in 8 32; do
RDPN6D_LIB=$R/tools/debug/lib_old.so python tools/grad_checksum.py $b > $O/sum_old_$b.txt 2>$O/err.txt
python tools/grad_checksum.py $b > $O/sum_new_$b.txt 2>>$O/err.txt
diff $O/sum_old_$b.txt $O/sum_new_$b.txt && echo "B=$b identical"; cat $O/sum_new_$b.txt
done
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "generalised or maxpool or deterministic or sym" > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/proft -- python3 $R/bench.py --train --dtype bf16 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench.json 2>/dev/null
cd $R; f=$(ls $O/proft/*/*kernel_trace.csv | head -1); python3 tools/train_trace_summary.py $f 1 400 > $O/train_trace_summary.txt; rm -rf $O/proft
grep -n "maxpool3x3s2_bwd\|dense_loss_k\|global_max_concat_bwd\|upsample_bilinear_bwd" $O/train_trace_summary.txt
