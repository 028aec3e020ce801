#!/bin/bash
# SQ / LDS / cache counters of the deformable kernels: gather form vs register-sampling window form
# on every layer shape of tools/bench_dcn2.py (B=32, real offset maps).  GPU box only; separate
# counter-only passes (kernel trace + --pmc).   bash tools/pmc_dcn.sh > gpurun_out/pmcd/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcd
mkdir -p $OUT
pass() {
  name=$1; shift
  KNOB=23 VALUES=1,2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_dcn2.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA
pass p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pass p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass p5 TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4", "p5"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dcn_reg_kernel" in k or ("igemm_kernel" in k and ", 2, false" in k):
                # one row per (kernel, grid): the seven layer shapes launch different grids
                key = (k.split("(")[0][-58:], r.get("Grid_Size", "?"))
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(p, k[0], "grid", k[1], {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
cat $OUT/fail.log 2>/dev/null; true
