#!/bin/bash
# SQ counters of the f32s LDS-halo kernel (64->64@128^2 and 128->128@64^2, B = 32), full kernel and
# the staging-free ablation.  GPU box only; counters + kernel trace only, one group per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcf
rm -rf $OUT; mkdir -p $OUT
pass() {
  name=$1; modes=$2; shift 2
  ONLY=0,2 MODES=$modes timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/ablate_halo.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 0 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass p2 0 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA
pass p3 0 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass p4 0 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass q1 7 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass q3 7 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for p in ("p1","p2","p3","p4","q1","q3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3x3s1" in k:
                agg[k.split("(")[0][-80:] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(p, k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
cat $OUT/summary.txt; cat $OUT/fail.log 2>/dev/null; tail -3 $OUT/p1.log
