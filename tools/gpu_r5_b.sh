#!/bin/bash
# round 5, session B: where does a wave-step of the team kernel wait?  probes: L1 weight stream, global path, offsets
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_dcn.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "benchmark_batch or fixtures" 2>&1 | tail -3 > $O/pytest_b.log; tail -2 $O/pytest_b.log
FORM=4 DBG=0,2,4,6,8,14 B=32 timeout 300 python tools/bench_dcn2.py > $O/team_probes_real.txt 2>&1; cat $O/team_probes_real.txt
ZERO_OFF=1 FORM=4 DBG=0,2,8,10 B=32 timeout 300 python tools/bench_dcn2.py > $O/team_probes_zero.txt 2>&1; cat $O/team_probes_zero.txt
KNOB=23 VALUES=2,4 B=32 timeout 300 python tools/bench_dcn2.py > $O/forms_coalesced.txt 2>&1; cat $O/forms_coalesced.txt
