#!/bin/bash
# mid-round GPU session: full GPU suite, smoke, the profile set of config 1, the driver's default command,
# one bench line for config 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
RND=${1:-r04}; TAG=${2:-v3}
O=gpurun_out/mid; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh $RND $TAG 1 > $O/prof1.log 2>&1; tail -12 $O/prof1.log | cut -c1-200
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-200 $O/bench_cfg2.json
