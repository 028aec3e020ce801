cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s13; mkdir -p $O
SKIPCHECK=1 PROBE=1 ROUNDS=1 timeout 300 python tools/bench_c3p.py > $O/c3p_probe.txt 2>&1; grep -A1 "res [01] | full" $O/c3p_probe.txt | cut -c1-420
