"""One batch on one stream vs the same batch split over S HIP streams (tail / ramp overlap).
python tools/bench_streams.py [arch] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth
from centernet_amd.model import create_model
arch = sys.argv[1] if len(sys.argv) > 1 else "resdcn_18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
m = create_model(arch, {"hm": 80, "wh": 2, "reg": 2}, 256 if arch.startswith("dla") else 64)
synth.fill_state_dict_(m, 317)
m = m.to(dev).eval()
x = synth.images(B, 512, 512, 0).to(dev)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    plan = m.plan_for(B, 512, 512, dev)
    base = timed(lambda: plan.run(x))
    print("%s B=%d one stream      %.3f ms  %.0f img/s" % (arch, B, base, B / base * 1e3))
    for S in (2, 4):
        xs = [c.contiguous() for c in x.chunk(S)]
        plans = []
        for xi in xs:
            m.invalidate_plans()
            plans.append(m.plan_for(xi.shape[0], 512, 512, dev))   # separate activation buffers per stream
        streams = [torch.cuda.Stream() for _ in range(S)]

        def run():
            cur = torch.cuda.current_stream()
            for s, p, xi in zip(streams, plans, xs):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    p.run(xi)
            for s in streams:
                cur.wait_stream(s)
        ms = timed(run)
        print("%s B=%d %d streams x %d   %.3f ms  %.0f img/s" % (arch, B, S, B // S, ms, B / ms * 1e3))
