"""Debug driver: one persistent-kernel launch per case, flushed prints (GPU box)."""
import os, sys
os.environ.setdefault("QUICK", "1")
sys.argv = [sys.argv[0]]
import importlib.util, builtins
src = open(os.path.join(os.path.dirname(__file__), "bench_c3p.py")).read()
src = src.split("allok = True")[0]
exec(compile(src, "bench_c3p_head", "exec"))
import torch
for key in (2,):
    for kw in (dict(B=1, ci=64, H=20, W=24, co=64), dict(B=2, ci=128, H=16, W=32, co=128, res=2),
               dict(B=2, ci=64, H=16, W=16, co=128, out_plain=True, relu=False), dict(B=2, ci=80, H=16, W=16, co=64, res=1, out_plain=True),
               dict(B=9, ci=64, H=64, W=64, co=64, res=1), dict(B=32, ci=64, H=128, W=128, co=64, res=1), dict(B=32, ci=512, H=16, W=16, co=512, res=1)):
        for knobs in (0,):
            lib.cn_set_tuning(28, key)
            c = Case(**kw)
            print("launch", key, kw, flush=True)
            c.launch()
            torch.cuda.synchronize()
            print("  done", float(c.y.abs().sum()), flush=True)
