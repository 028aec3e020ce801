#!/bin/bash
# SQ counters of every kernel of the benchmark step (matrix-pipe busy, wait share, instruction mix):
#   bash tools/pmc_step.sh [config]      -> gpurun_out/pmc_step/sq_counters_cfgN.txt
# rocprofv3 --kernel-trace --pmc, one counter group per pass (no other trace domain).  GPU box only.
CFG=${1:-1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_step
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-secondary"
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BENCH > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass p2 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU
python - > $OUT/sq_counters_cfg$CFG.txt <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# SQ counters per kernel of: bench.py --config $CFG (mean over launches; rocprofv3 --pmc, two passes)")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
rows = []
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc else 0
    wait = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0
    rows.append((cyc * n, k, n, cyc, busy, wait, m))
for tot, k, n, cyc, busy, wait, m in sorted(rows, reverse=True)[:24]:
    print("%-78s n=%4d  cycles/launch %9.0f  mfma_busy %5.1f %%  wait %5.1f %%  valu/mfma %5.1f" % (
        k[:78], n, cyc, 100 * busy, 100 * wait, m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_INSTS_MFMA", 0), 1)))
PY
rm -rf $OUT/p1 $OUT/p2
cat $OUT/sq_counters_cfg$CFG.txt | cut -c1-200; cat $OUT/fail.log 2>/dev/null
