#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "window or benchmark_batch or deformable" 2>&1 | tail -4 > $O/pytest.log; tail -3 $O/pytest.log
KNOB=23 VALUES=2,4,5 B=32 timeout 300 python tools/bench_dcn2.py > $O/dcn_forms.txt 2>&1; cat $O/dcn_forms.txt
FORM=4 DBG=0,8,128 B=32 timeout 300 python tools/bench_dcn2.py > $O/dcn_team_probes.txt 2>&1; cat $O/dcn_team_probes.txt
for t in 0 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-secondary --tune 36=$t > $O/bench_team$t.json 2> $O/bench_team$t.err
  python - <<PY
import json
try:
    r=json.load(open("$O/bench_team$t.json"))
    print("team=$t", round(r["value"],1), "img/s dcn", round(r["roofline_dcn_mfma"]["frac"],4), round(r["roofline_dcn_mfma"]["avg_launch_ms"],4), "conv", round(r["roofline"]["frac"],4))
except Exception as e:
    print("team=$t failed", e); print(open("$O/bench_team$t.err").read()[-600:])
PY
done
