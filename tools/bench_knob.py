"""Whole-network A/B of one tuning knob: python tools/bench_knob.py KEY V0 V1 [arch] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.model import create_model
key, v0, v1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
arch = sys.argv[4] if len(sys.argv) > 4 else "resdcn_18"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
lib = native.lib()
dev = torch.device("cuda:0")
m = create_model(arch, {"hm": 80, "wh": 2, "reg": 2}, 256 if arch.startswith("dla") else 64)
synth.fill_state_dict_(m, 317)
m = m.to(dev).eval()
if os.environ.get('FP16'):
    m.half_compute()
x = synth.images(B, 512, 512, 0).to(dev)
res = {}
for rnd in range(3):
    for v in (v0, v1):
        lib.cn_set_tuning(key, v)
        m.invalidate_plans()
        with torch.no_grad():
            for _ in range(3): m(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10): m(x)
            torch.cuda.synchronize()
        res.setdefault(v, []).append((time.perf_counter() - t) / 10 * 1e3)
for v, ts in res.items():
    print("key %d = %d : %s ms/forward (min %.3f)" % (key, v, " ".join("%.3f" % t for t in ts), min(ts)))

