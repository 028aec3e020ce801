cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_detector.py tests/test_gpu_conv.py tests/test_gpu_pre.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "run_frames or stem or pre" 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 400 python tools/bench_e2e.py > $O/e2e.txt 2>&1; tail -7 $O/e2e.txt | cut -c1-330
