#!/bin/bash
# end-of-round GPU session: full GPU suite, then the profile sets of configs 1 and 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh ${1:-r03} ${2:-v2} 1 > $O/prof1.log 2>&1; tail -30 $O/prof1.log | cut -c1-300
bash tools/profile_round.sh ${1:-r03} ${2:-v2} 2 > $O/prof2.log 2>&1; tail -30 $O/prof2.log | cut -c1-300
