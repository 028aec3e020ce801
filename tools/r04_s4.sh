cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "heads or recalib or far_below" 2>&1 | tail -5 > $O/pytest_heads.log; cat $O/pytest_heads.log
SKIPCHECK=1 KNOBS=0,8,10,12,14 timeout 300 python tools/bench_c3p.py 2>&1 | grep "res " | cut -c1-330 > $O/c3p_resat.txt; cat $O/c3p_resat.txt
for t in 1 0; do
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 31=$t --per-op > $O/bench_h$t.json 2> $O/bench_h$t.perop; cut -c1-200 $O/bench_h$t.json; grep "heads" $O/bench_h$t.perop
done
