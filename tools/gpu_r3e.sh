#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dcn.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "window" 2>&1 | tail -15 > $O/pytest_win.log
tail -6 $O/pytest_win.log
KNOB=23 VALUES=1,0 timeout 300 python tools/bench_dcn2.py > $O/dcn_form.txt 2>&1
cat $O/dcn_form.txt
timeout 600 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_f32s_range.py tests/test_gpu_net.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.log
tail -4 $O/pytest.log
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 23=1 > $O/bench_g.json 2> $O/bench_g.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_w.json 2> $O/bench_w.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 --tune 23=1 > $O/bench_dla_g.json 2> $O/bench_dla_g.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 > $O/bench_dla_w.json 2> $O/bench_dla_w.err
for f in bench_g bench_w bench_dla_g bench_dla_w; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['time_share'], d['roofline_dcn_mfma']['frac'])"; done
