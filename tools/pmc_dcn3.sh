#!/bin/bash
# SQ / LDS counters of the deformable window kernels (register-sampling form vs team form) on two layer
# shapes of tools/bench_dcn2.py (B = 32, real offset maps).  GPU box only; separate counter-only passes.
#   FORMS=2,4 bash tools/pmc_dcn3.sh > gpurun_out/pmcd3/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcd3
mkdir -p $OUT
FORMS=${FORMS:-2,4}
pass() {
  name=$1; shift
  ONLY=${ONLY:-2,3} KNOB=23 VALUES=$FORMS timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_dcn2.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA
pass p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pass p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dcn_reg_kernel" in k or "dcn_team_kernel" in k or "dcn_trio_kernel" in k:
                key = (k.split("(")[0][-48:], r.get("Grid_Size", "?"))
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(p, k[0], "grid", k[1], {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
cat $OUT/fail.log 2>/dev/null; true
