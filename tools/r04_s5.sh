cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_detector.py tests/test_gpu_ddd.py tests/test_gpu_exct.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_decode.log; cat $O/pytest_decode.log
timeout 200 python tools/bench_decode.py > $O/bench_decode.txt 2>&1; tail -12 $O/bench_decode.txt | cut -c1-200
SKIPCHECK=1 KNOBS=8,10,12,14 ROUNDS=7 timeout 300 python tools/bench_c3p.py 2>&1 | grep "res 1" | cut -c1-330 > $O/c3p_resat.txt; cat $O/c3p_resat.txt
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --per-op > $O/bench.json 2> $O/bench.perop; cut -c1-150 $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/s5/bench.json'))
for k in ('roofline_decode_hbm','box_calibration','headline_over_calibration','range_tracking_off_leg','fp32_mfma_leg','time_share'):
    print(k, json.dumps(d.get(k))[:400])
PY
