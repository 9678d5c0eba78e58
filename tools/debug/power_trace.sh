#!/bin/bash
# board power / shader clock sampled while the headline bench runs sustained: bash tools/debug/power_trace.sh <tag> [bench args]
O=gpurun_out/${1:-power}; mkdir -p $O; shift
rocm-smi --showpower --showclocks --showmaxpower > $O/idle.txt 2>&1
python bench.py --no-cpu-baseline --steps 3000 "$@" > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12   # model build + warm-up
for i in $(seq 1 24); do
  echo "--- sample $i" >> $O/samples.txt
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" >> $O/samples.txt
  sleep 0.5
done
wait $BP
cut -c1-160 $O/bench.json
grep -E "Max|Power" $O/idle.txt | head -5
grep -E "Power|sclk" $O/samples.txt | sort | uniq -c | sort -rn | head -12
