#!/bin/bash
# full GPU suite + smoke + the driver's default bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
r=json.load(open("$O/bench_default.json"))
print(round(r["value"],1), "img/s", round(r["ms_per_step"],3), "ms; conv", round(r["roofline"]["frac"],3), "dcn", round(r["roofline_dcn_mfma"]["frac"],3), "decode", round(r["roofline_decode_hbm"]["frac"],3))
for k,v in r.get("secondary_configs",{}).items():
    print(k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","error","range_clean")})
PY
