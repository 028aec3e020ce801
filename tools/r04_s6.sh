cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_decode.log; cat $O/pytest_decode.log
timeout 200 python tools/bench_decode.py > $O/bench_decode.txt 2>&1; tail -12 $O/bench_decode.txt | cut -c1-200
