#!/bin/bash
# bash tools/knob_sweep.sh "<bench flags>" k=v [k=v ...]: bench line per single-knob setting (and the default, twice)
flags=$1; shift
run() { python bench.py --steps 30 --warmup 10 --no-cpu-baseline $flags $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$1', round(d['value'],1), round(d['ms_per_step'],3), d['time_share'])"; }
run default ""
for kv in "$@"; do run $kv "--tune $kv"; done
run default ""
