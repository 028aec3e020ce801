#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dcn.py -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -15 > $O/pytest_dcn.log
tail -3 $O/pytest_dcn.log
KNOB=23 VALUES=2,1,2 timeout 300 python tools/bench_dcn2.py > $O/dcn_form.txt 2>&1
cat $O/dcn_form.txt
timeout 1200 python -m pytest tests/test_gpu_f32s_range.py tests/test_gpu_net.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.log
tail -3 $O/pytest.log
for cfg in 1 2; do for form in 1 0; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --config $cfg --tune 23=$form > $O/bench_c${cfg}_f${form}.json 2> $O/bench_c${cfg}_f${form}.err
python -c "
import json,sys
d=json.loads(open('$O/bench_c${cfg}_f${form}.json').read().strip().splitlines()[-1]); print('cfg$cfg form$form', round(d['value']), d.get('time_share'), d.get('roofline_dcn_mfma',{}).get('frac'))"
done; done
