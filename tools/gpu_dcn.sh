#!/bin/bash
# the LDS-window deformable forms (cn_dcn2 / 3 / 4.hip): their tests, then the layer shapes under every form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/dcn; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dcn.py -q -x --timeout 600 -p no:cacheprovider -k "window or benchmark_batch" 2>&1 | tail -15 > $O/pytest_dcn.log
tail -5 $O/pytest_dcn.log
KNOB=23 VALUES=${VALUES:-0,4,5,6,7} timeout 600 python tools/bench_dcn2.py > $O/bench_dcn2.txt 2>&1
cat $O/bench_dcn2.txt
