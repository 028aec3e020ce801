#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3b; mkdir -p $O; rm -f gpurun_out/f32s_sweep.jsonl gpurun_out/parity_fractions.jsonl
python -m pytest tests/test_gpu_f32s_range.py tests/test_gpu_dcn.py tests/test_gpu_net.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -60 > $O/pytest.log
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
CN_RANGE=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_norange.json 2> $O/bench_norange.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench2.json 2> $O/bench2.err
CN_RANGE=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_norange2.json 2> $O/bench_norange2.err
tail -5 $O/pytest.log
for f in bench bench_norange bench2 bench_norange2; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['time_share'], d.get('fp32_mfma_leg'))"; done
