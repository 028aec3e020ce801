#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pipe; mkdir -p $O
timeout 300 tools/ubench/mfma_operand_hazard > $O/ubench_hazard.txt 2>&1; grep -v ":     0 of" $O/ubench_hazard.txt | head -40; echo "lines with zero hits: $(grep -c ':     0 of' $O/ubench_hazard.txt) of $(wc -l < $O/ubench_hazard.txt)"
SKIPCHECK=1 KNOBS=2 STAG=0,4,8,16 ROUNDS=5 timeout 600 python tools/bench_c3p.py > $O/bench_c3p_stag2.txt 2>&1; tail -10 $O/bench_c3p_stag2.txt
for sg in 64 0 8 16 0 64; do
timeout 600 python bench.py --no-secondary --tune 30=2 --tune 29=$sg --steps 30 > $O/bench_s$sg.json 2> $O/bench.err; python -c "
import json; r=json.load(open('$O/bench_s$sg.json')); print('knobs 2 stagger $sg', round(r['value'],1), round(r['ms_per_step'],4), round(r['roofline']['frac'],4))"
done
