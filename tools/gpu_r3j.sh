#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dcn.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "window or f32s" 2>&1 | tail -15 > $O/pytest_dcn.log
tail -5 $O/pytest_dcn.log
KNOB=23 VALUES=2,1,2 timeout 300 python tools/bench_dcn2.py > $O/dcn_form.txt 2>&1
cat $O/dcn_form.txt
DBG=0,8,128 timeout 300 python tools/bench_dcn2.py > $O/dcn_dbg.txt 2>&1
cat $O/dcn_dbg.txt
