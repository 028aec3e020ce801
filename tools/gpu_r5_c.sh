#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c; mkdir -p $O
TRACE=1 ONLY=2,3,1 FORM=4 DBG=0,512 B=32 timeout 300 python tools/bench_dcn2.py > $O/team_trace.txt 2>&1; cat $O/team_trace.txt
