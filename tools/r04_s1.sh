cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s1; mkdir -p $O
timeout 300 python tools/bench_c3p.py > $O/c3p.txt 2>&1; tail -25 $O/c3p.txt | cut -c1-260
for t in 1 0; do
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 28=$t --per-op > $O/bench_c3p$t.json 2> $O/bench_c3p$t.perop; cut -c1-300 $O/bench_c3p$t.json
done
timeout 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest_conv.log; cat $O/pytest_conv.log
