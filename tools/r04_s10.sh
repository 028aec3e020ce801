cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_detector.py tests/test_gpu_conv.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "decode or topk or select or detector or run_frames or stem or pool" 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
timeout 200 python tools/bench_decode.py > $O/bench_decode.txt 2>&1; grep -A3 "heat map" $O/bench_decode.txt | cut -c1-200
for e in 1 0; do
CN_STEM_Y_F32S=$e timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --per-op > $O/bench_y$e.json 2> $O/bench_y$e.perop; cut -c1-150 $O/bench_y$e.json; grep -E "^op  [0-2] " $O/bench_y$e.perop
done
timeout 300 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-180 $O/bench_cfg4.json
