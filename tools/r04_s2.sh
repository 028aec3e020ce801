cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_f32s_range.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "heads or range or recalib or calibr or far_below or clamped" 2>&1 | tail -15 > $O/pytest_heads.log; cat $O/pytest_heads.log
for t in 1 0; do
timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --tune 31=$t --per-op > $O/bench_h$t.json 2> $O/bench_h$t.perop; cut -c1-200 $O/bench_h$t.json; grep "heads" $O/bench_h$t.perop
done
