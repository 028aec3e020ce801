#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5d; mkdir -p $O
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_f32s_range.py tests/test_gpu_conv.py -m gpu -q --timeout 600 -p no:cacheprovider -k "window or benchmark_batch or deformable or offset_conv or module" 2>&1 | tail -3; done
KNOB=23 VALUES=2,4,5 B=32 timeout 300 python tools/bench_dcn2.py > $O/dcn_forms.txt 2>&1; cat $O/dcn_forms.txt
N=6 timeout 300 python tools/dbg_dcn_team.py 2>&1 | grep -c "bad \[\]"
