#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/exp_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/exp_tests.log
tail -4 gpurun_out/exp_tests.log
B=32 timeout 300 python tools/bench_dcn.py > gpurun_out/exp_dcn.log 2>&1; cat gpurun_out/exp_dcn.log
CN_FUSE_HEADS=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-op > gpurun_out/exp_bench_nofuse.json 2> gpurun_out/exp_bench_nofuse.err
CN_FUSE_HEADS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-op > gpurun_out/exp_bench_fuse.json 2> gpurun_out/exp_bench_fuse.err
grep "^op" gpurun_out/exp_bench_nofuse.err | tail -4
grep "^op" gpurun_out/exp_bench_fuse.err | tail -12
python -c "
import json
for f in ('nofuse','fuse'):
    d=json.load(open('gpurun_out/exp_bench_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'])
"
