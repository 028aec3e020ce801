#!/bin/bash
# the stem kernels (cn_stem.hip): their tests, then the per-op listing of config 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/stem; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_f32s_range.py -q -x --timeout 600 -p no:cacheprovider -k "stem" 2>&1 | tail -15 > $O/pytest_stem.log
tail -5 $O/pytest_stem.log
python bench.py --config 1 --per-op --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/per_op.txt
grep "^op" $O/per_op.txt | head -8
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_dcn_mfma"]["frac"])
PY
