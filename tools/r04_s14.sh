cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s14; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "concat or dla or stem or chunks or maxpool or pool" 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
for e in 1 0; do
CN_CONCAT_INPLACE=$e timeout 300 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_cfg2_$e.json 2> $O/err_$e.txt; cut -c1-150 $O/bench_cfg2_$e.json
python - <<PY
import json
d=json.load(open('gpurun_out/s14/bench_cfg2_$e.json')); print('  time_share', d['time_share'])
PY
done
