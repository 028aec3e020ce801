"""Eager launch list vs HIP-graph replay of the whole forward (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth
from centernet_amd.model import create_model
arch = sys.argv[1] if len(sys.argv) > 1 else "resdcn_18"
dev = torch.device("cuda:0")
m = create_model(arch, {"hm": 80, "wh": 2, "reg": 2}, 256 if arch.startswith("dla") else 64)
synth.fill_state_dict_(m, 317)
m = m.to(dev).eval()
if os.environ.get("FP16") == "1":       # BASELINE configs[4]: hourglass fp16
    m.half_compute(True)
for B in [int(b) for b in os.environ.get("B", "1,4,32").split(",")]:
    x = synth.images(B, 512, 512, 0).to(dev)
    plan = m.plan_for(B, 512, 512, dev)
    def t(n=20):
        for _ in range(3): plan.run(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): plan.run(x)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = t()
    plan.capture()
    graph = t()
    print("launches per forward: %d" % len(plan.b.ops))
    print("%s B=%2d  eager %.3f ms  graph %.3f ms  (%.0f vs %.0f img/s)" % (arch, B, eager, graph, B / eager * 1e3, B / graph * 1e3))
    m.invalidate_plans()
