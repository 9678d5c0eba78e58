#!/bin/bash
# The C oracles under AddressSanitizer + UBSan (CPU build only - GPU sanitizers are not available on this pool): builds an
# instrumented oracle/liboracle.so, runs the oracle test files against it, restores the normal build.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
gcc -O1 -g -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -shared -o oracle/liboracle.so oracle/fps_oracle.c oracle/ransac_oracle.c oracle/pnp_oracle.c -lm
trap 'rm -f oracle/liboracle.so; make -s -C oracle liboracle.so' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_fps_oracle.py tests/test_ransac_oracle.py tests/test_pnp_oracle.py tests/test_select_oracle.py -x -q
