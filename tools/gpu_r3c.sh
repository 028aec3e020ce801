#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c; mkdir -p $O; rm -f gpurun_out/f32s_sweep.jsonl gpurun_out/parity_fractions.jsonl
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
CN_RANGE=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_norange.json 2> $O/bench_norange.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench2.json 2> $O/bench2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/s -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/s -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/s
tail -5 $O/pytest.log
for f in bench bench_norange bench2; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['time_share'], d.get('fp32_mfma_leg'), d.get('dominant_launch'))"; done
head -12 $O/kernel_stats.csv | cut -c1-220
