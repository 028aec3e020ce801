#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3h; mkdir -p $O
DBG=0,2,4,8,64,128,14,78 timeout 300 python tools/bench_dcn2.py > $O/dcn_dbg.txt 2>&1
cat $O/dcn_dbg.txt
ZERO_OFF=1 DBG=0,1 timeout 300 python tools/bench_dcn2.py > $O/dcn_zero.txt 2>&1
cat $O/dcn_zero.txt
