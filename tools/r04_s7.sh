cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_conv.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "decode or topk or transpose or select" 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
timeout 200 python tools/bench_decode.py > $O/bench_decode.txt 2>&1; grep -A3 "heat map" $O/bench_decode.txt | cut -c1-200
for t in 1 0; do
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 32=$t --per-op > $O/bench_d$t.json 2> $O/bench_d$t.perop; cut -c1-150 $O/bench_d$t.json; grep "deconv" $O/bench_d$t.perop
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/s7/bench_d1.json'))
for k in ('roofline_decode_hbm','time_share'):
    print(k, json.dumps(d.get(k))[:300])
PY
