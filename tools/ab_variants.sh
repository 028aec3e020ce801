#!/bin/bash
# A/B of library variants built by tools/build_variant.sh, on the GPU box:
#   bash tools/ab_variants.sh <out_tag> <variant> [<variant> ...]
# writes gpurun_out/ab_<tag>_<variant>.{f32s.txt,bench.json,perop.txt}
tag=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  export CENTERNET_AMD_LIB=$PWD/centernet_amd/variants/libcenternet_amd_$v.so
  python tools/bench_f32s.py > gpurun_out/ab_${tag}_$v.f32s.txt 2>&1
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/ab_${tag}_$v.bench.json 2> gpurun_out/ab_${tag}_$v.err
  python bench.py --steps 10 --warmup 5 --no-cpu-baseline --per-op > /dev/null 2> gpurun_out/ab_${tag}_$v.perop.txt
done
unset CENTERNET_AMD_LIB
for v in "$@"; do echo "== $v"; tail -7 gpurun_out/ab_${tag}_$v.f32s.txt; python - <<PY
import json
for l in open("gpurun_out/ab_${tag}_$v.bench.json"):
    if l.startswith("{"):
        d = json.loads(l); print("bench", d["value"], d["ms_per_step"])
PY
done
