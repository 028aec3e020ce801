#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dcn.py -m gpu -q --timeout 300 -p no:cacheprovider -k "window" 2>&1 | tail -25 > $O/pytest_win.log
tail -12 $O/pytest_win.log
KNOB=23 VALUES=1,2,3 timeout 300 python tools/bench_dcn2.py > $O/dcn_form.txt 2>&1
cat $O/dcn_form.txt
