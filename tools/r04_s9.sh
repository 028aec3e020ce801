cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in 2 3 4; do
timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; cut -c1-180 $O/bench_cfg$c.json
python - <<PY
import json
d=json.load(open('gpurun_out/s9/bench_cfg$c.json'))
print('  time_share', d['time_share'], 'conv frac', round(d['roofline']['frac'],3))
PY
done
