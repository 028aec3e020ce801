cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_detector.py -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
for b in 1 32; do for h in iid net floor; do
B=$b HEAT=$h timeout 120 python tools/bench_decode.py 2>/dev/null | grep -E "heat map|default|flag 2048" | tr '\n' ' ' | cut -c1-230; echo " [B=$b]"
done; done > $O/decode_sweep.txt; cat $O/decode_sweep.txt
timeout 400 python tools/bench_e2e.py > $O/e2e.txt 2>&1; tail -6 $O/e2e.txt | cut -c1-150
