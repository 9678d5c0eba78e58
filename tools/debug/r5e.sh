mkdir -p gpurun_out/r5_e; O=gpurun_out/r5_e
RDPN6D_STEM_V1=1 python tools/bench_stem.py 2>&1 | tail -3
for fl in "" 1; do
RDPN6D_STEM_V1=1 FLUSH=$fl python tools/bench_stem.py 2>&1 | tail -1
for sk in 0 64 128 192 256 320 448; do RDPN6D_STEM_SKEW=$sk FLUSH=$fl python tools/bench_stem.py 2>&1 | tail -1; done
done > $O/stem_skew.log; cat $O/stem_skew.log
