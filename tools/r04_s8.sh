cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "stride2 or transpose_4x4_s2_on or persistent" 2>&1 | tail -12 > $O/pytest.log; cat $O/pytest.log
for t in 1 0; do
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --tune 33=$t --per-op > $O/bench_s$t.json 2> $O/bench_s$t.perop; cut -c1-150 $O/bench_s$t.json; grep -E "op  6|op 11|op 16" $O/bench_s$t.perop
done
