#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_conv.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -4 > $O/pytest.log
tail -3 $O/pytest.log
for t in 0 1 3 0 1 3; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --per-op --tune 24=$t > $O/bench_t$t.json 2> $O/bench_t$t.err
python -c "
import json
d=json.loads(open('$O/bench_t$t.json').read().strip().splitlines()[-1]); print('key24=$t', round(d['value']), d['ms_per_step'])"
grep "^op 29\|^op 1[1-9] " $O/bench_t$t.err | awk '{printf \"%s:%s \", \$2, \$5}'; echo
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 --tune 24=0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dla t0', round(d['value']))"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg --config 2 --tune 24=3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dla t3', round(d['value']))"
