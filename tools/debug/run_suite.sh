#!/bin/bash
# the whole GPU suite + smoke, log under gpurun_out/<tag>/
O=gpurun_out/${1:-suite}; mkdir -p $O
python -m pytest tests -m gpu -q -rs --durations=15 > $O/gpu_tests.log 2>&1; tail -40 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
