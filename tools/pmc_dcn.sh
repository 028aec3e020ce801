#!/bin/bash
# PMC passes over the deformable kernel (tools/bench_dcn.py, B=32).  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcd
mkdir -p $OUT
pass() {
  name=$1; shift
  B=32 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_dcn.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/fail.log
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
pass p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA
pass p3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
pass p4 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass p5 TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass p6 TA_BUSY_avr TD_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM
cat $OUT/fail.log 2>/dev/null; true
