#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5e; mkdir -p $O
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-secondary --per-op > $O/b1.json 2> $O/b1_per_op.txt; grep "^op" $O/b1_per_op.txt
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-secondary --per-op > $O/b4.json 2> $O/b4_per_op.txt; grep "^op" $O/b4_per_op.txt | awk '{k=$3" "$4; ms[k]+=$5; n[k]++} END{for(k in ms) printf "%s n=%d %.3f ms\n", k, n[k], ms[k]}' | sort -k5 -n -r | head -50
python -c "
import json; r=json.load(open('$O/b4.json')); print(r['value'], r['ms_per_step'], r['time_share'])"
