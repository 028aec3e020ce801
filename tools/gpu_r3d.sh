#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3d; mkdir -p $O
python tools/bench_dcn2.py > $O/dcn_tile2d.txt 2>&1
python -m pytest tests/test_gpu_dcn.py tests/test_gpu_f32s_range.py -m gpu -q --timeout 900 -p no:cacheprovider -k "dcn or deformable" 2>&1 | tail -8 > $O/pytest.log
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 22=0 > $O/bench_t0.json 2> $O/bench_t0.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_t1.json 2> $O/bench_t1.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 --tune 22=0 > $O/bench_dla_t0.json 2> $O/bench_dla_t0.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 > $O/bench_dla_t1.json 2> $O/bench_dla_t1.err
cat $O/dcn_tile2d.txt; tail -3 $O/pytest.log
for f in bench_t0 bench_t1 bench_dla_t0 bench_dla_t1; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['time_share'], d['roofline_dcn_mfma']['frac'])"; done
