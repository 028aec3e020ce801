#!/bin/bash
# One GPU-box session producing the artefacts under profiles/ (run through gpurun):
#   bash tools/profile_round.sh r02 v1 [config]
# 1. rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters only + kernel trace) -> pmc_traffic json
# 2. python bench.py (same flags) -> bench json (+ per-op listing)
# 3. rocprofv3 --kernel-trace --stats of the same command -> kernel stats csv
ROUND=${1:-r03}
TAG=${2:-v1}
CFG=${3:-1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${ROUND}_${TAG}_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 30 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-secondary"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- $BENCH > $OUT/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- $BENCH > $OUT/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $OUT/f $OUT/w $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
cp $OUT/pmc_traffic.json $R/profiles/${ROUND}_pmc_traffic_cfg$CFG.json   # so that step 2 reports this traffic
cd $R
timeout 900 python bench.py --config $CFG --steps 50 --warmup 5 --per-op --no-secondary > $OUT/bench.json 2> $OUT/bench_per_op.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o p -- python $R/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
cp $OUT/s/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/s -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/f $OUT/w $OUT/s
cat $OUT/bench.json | cut -c1-600; head -14 $OUT/kernel_stats.csv | cut -c1-200; tail -25 $OUT/pmc_traffic.txt
