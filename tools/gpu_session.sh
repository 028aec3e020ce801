#!/bin/bash
# ONE GPU-box session (the only session script: per-round copies are not kept): [full GPU suite,] smoke, the profile
# sets of configs 1 and 2, the driver's default command, one bench line each for configs 3 and 4, host-boundary
# rates, SQ counters of the step, decode forms
#   bash tools/gpu_session.sh r06 v1 [nosuite | suiteonly]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
RND=${1:-r06}; TAG=${2:-v1}
O=gpurun_out/final; mkdir -p $O
if [ "$3" != "nosuite" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
if [ "$3" == "suiteonly" ]; then
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
  exit 0
fi
bash tools/profile_round.sh $RND $TAG 1 > $O/prof1.log 2>&1; tail -12 $O/prof1.log | cut -c1-200
bash tools/profile_round.sh $RND $TAG 2 > $O/prof2.log 2>&1; tail -12 $O/prof2.log | cut -c1-200
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-200 $O/bench_cfg3.json
python bench.py --config 4 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-200 $O/bench_cfg4.json
timeout 300 python tools/bench_e2e.py > $O/e2e.txt 2>&1; tail -6 $O/e2e.txt
bash tools/pmc_step.sh 1 > $O/pmc_step1.log 2>&1; tail -3 $O/pmc_step1.log | cut -c1-160
for h in iid net floor; do HEAT=$h timeout 100 python tools/bench_decode.py 2>&1 | grep -v amdgpu.ids; done > $O/decode_forms.txt; grep "default product" $O/decode_forms.txt
