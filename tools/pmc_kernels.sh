#!/bin/bash
# PMC passes over tools/bench_kernels.py (two shapes, generic vs halo kernel).  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmck
mkdir -p $OUT
pass() {
  name=$1; shift
  ONLY=0,4 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_kernels.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA
pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
python - <<PY
import csv, glob, collections
for p in ("p1","p2","p3","p4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3x3s1" in k or "igemm" in k:
                agg[k.split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(p, k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
cat $OUT/fail.log 2>/dev/null; tail -3 $OUT/p1.log
