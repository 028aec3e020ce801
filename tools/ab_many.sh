#!/bin/bash
# bash tools/ab_many.sh "<bench flags>" <variant> [<variant> ...]: bench line of the in-tree library and of each variant build
flags=$1; shift
for v in intree "$@"; do
  if [ "$v" = intree ]; then unset CENTERNET_AMD_LIB; else export CENTERNET_AMD_LIB=$PWD/centernet_amd/variants/libcenternet_amd_$v.so; fi
  for rep in 1 2; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline $flags 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
