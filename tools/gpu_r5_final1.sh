#!/bin/bash
# round-5 profile set on the final tree: rocprof kernel stats + PMC traffic + per-op listing of configs[1], per-op listings at B = 1 / 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_round.sh r05 v3 1 > gpurun_out/prof_r05_v3.log 2>&1; tail -30 gpurun_out/prof_r05_v3.log | cut -c1-220
O=gpurun_out/small; mkdir -p $O
for b in 1 4; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-fp32-leg --batch $b --steps 50 --per-op > $O/bench_b$b.json 2> $O/per_op_b$b.txt; python -c "
import json; r=json.load(open('$O/bench_b$b.json')); print('batch $b', round(r['value'],1), round(r['ms_per_step'],4))"
done
grep "^op" $O/per_op_b1.txt | cut -c1-100
