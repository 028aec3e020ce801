for t in "" "--tune 15=0" "--tune 21=2" "--tune 15=0 --tune 21=2"; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value']), round(d['ms_per_step'],3))"
done
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --per-op --tune 15=0 --tune 21=2 2>&1 >/dev/null | grep "^op" > gpurun_out/perop_t2.txt
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --per-op 2>&1 >/dev/null | grep "^op" > gpurun_out/perop_t0.txt
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
