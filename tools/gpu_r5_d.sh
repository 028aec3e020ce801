#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "(window or benchmark_batch or deformable) and 6" 2>&1 | tail -4
KNOB=23 VALUES=4,6 B=32 timeout 300 python tools/bench_dcn2.py > $O/dcn_forms.txt 2>&1; cat $O/dcn_forms.txt
bash tools/gpu_r5_c.sh
