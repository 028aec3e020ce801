#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3l; mkdir -p $O
for B in 32 16 8; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --batch $B --per-op > $O/bench_b$B.json 2> $O/bench_b$B.err
python -c "
import json
d=json.loads(open('$O/bench_b$B.json').read().strip().splitlines()[-1]); print('B=$B', round(d['value']), d['ms_per_step'], d.get('time_share'))"
done
grep "^op" $O/bench_b32.err > $O/perop_b32.txt; grep "^op" $O/bench_b16.err > $O/perop_b16.txt; grep "^op" $O/bench_b8.err > $O/perop_b8.txt
paste <(awk '{print $2,$3,$4,$5}' $O/perop_b32.txt) <(awk '{print $5}' $O/perop_b16.txt) <(awk '{print $5}' $O/perop_b8.txt) | head -60
