cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s15; mkdir -p $O
for e in 0 1; do
CN_DLA_LEVEL_PLAIN=$e timeout 300 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_cfg2_$e.json 2> $O/err_$e.txt; cut -c1-150 $O/bench_cfg2_$e.json
python - <<PY
import json
d=json.load(open('gpurun_out/s15/bench_cfg2_$e.json')); print('  time_share', d['time_share'])
PY
done
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "dla" 2>&1 | tail -3
