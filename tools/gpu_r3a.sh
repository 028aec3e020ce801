#!/bin/bash
# round-3 GPU session A: full GPU suite (no -x), bench lines with / without range tracking
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a; mkdir -p $O; rm -f gpurun_out/f32s_sweep.jsonl gpurun_out/parity_fractions.jsonl
python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -120 > $O/pytest.log
python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
CN_RANGE=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg > $O/bench_norange.json 2> $O/bench_norange.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-fp32-leg --per-op > $O/bench_perop.json 2> $O/bench_perop.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -5 $O/pytest.log; head -c 600 $O/bench.json; echo; tail -3 $O/smoke.log
