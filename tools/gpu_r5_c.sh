#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c; mkdir -p $O
KNOB=38 VALUES=0,8,16,32,64,128 B=32 timeout 300 python tools/bench_dcn2.py > $O/stagger.txt 2>&1; cat $O/stagger.txt
