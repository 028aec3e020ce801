#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "offset_conv" 2>&1 | tail -3 > $O/pytest.log; tail -2 $O/pytest.log
for t in "39=0" "40=0" "40=768" "40=100000"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-secondary --per-op --tune $t > $O/bench_oc.json 2> $O/bench_oc.err
  echo "$t: $(grep '^op' $O/bench_oc.err | awk '$2==20||$2==23||$2==26{printf "%s %s %s %s | ", $2,$3,$5,$6} END{print ""}')"
done
for t in "39=0" "40=0" "40=768"; do
timeout 400 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --tune $t > $O/bench_cfg2_oc.json 2> $O/bench_cfg2_oc.err; echo "$t $(cut -c1-110 $O/bench_cfg2_oc.json)"
done
