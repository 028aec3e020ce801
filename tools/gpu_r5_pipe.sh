#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pipe2; mkdir -p $O
V=$PWD/centernet_amd/variants/libcn_sched2.so
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "pipelined" -p no:cacheprovider 2>&1 | tail -3
SKIPCHECK=1 KNOBS=0,2 STAG=0 ROUNDS=7 EQ=40 timeout 600 python tools/bench_c3p.py > $O/c3p_sched1.txt 2>&1; grep "^(\|differ" $O/c3p_sched1.txt
echo "--- schedule 2"
CENTERNET_AMD_LIB=$V CHECK_KNOBS=2 KNOBS=0,2 STAG=0 ROUNDS=7 EQ=40 timeout 600 python tools/bench_c3p.py > $O/c3p_sched2.txt 2>&1; grep "^(\|differ\|FAIL\|ALL OK" $O/c3p_sched2.txt
for i in 1 2; do
timeout 600 python bench.py --no-secondary --steps 30 > $O/bench_s1.json 2> $O/bench.err; python -c "
import json; r=json.load(open('$O/bench_s1.json')); print('sched 1', round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4))"
CENTERNET_AMD_LIB=$V timeout 600 python bench.py --no-secondary --steps 30 > $O/bench_s2.json 2> $O/bench.err; python -c "
import json; r=json.load(open('$O/bench_s2.json')); print('sched 2', round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4))"
done
for b in 1 4; do for k in 1 2; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-fp32-leg --batch $b --steps 50 --tune 28=$k > $O/bench_b${b}_k$k.json 2> $O/bench.err; python -c "
import json; r=json.load(open('$O/bench_b${b}_k$k.json')); print('batch $b key28 $k', round(r['value'],1), round(r['ms_per_step'],4))"
done; done
