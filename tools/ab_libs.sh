#!/bin/bash
# bash tools/ab_libs.sh <variant> [bench flags]: bench + per-op of the in-tree library and of one variant build
v=$1; shift
for lib in "" "$PWD/centernet_amd/variants/libcenternet_amd_$v.so"; do
  export CENTERNET_AMD_LIB=$lib
  tag=${lib:+$v}; tag=${tag:-intree}
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],3))"
  python bench.py --steps 10 --warmup 5 --no-cpu-baseline --per-op "$@" 2>&1 >/dev/null | grep "^op" > gpurun_out/perop_$tag.txt
done
