"""Print the launch list of a network plan with tensor formats (GPU box): python tools/plan_dump.py dla_34 [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth
from centernet_amd.model import create_model
arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
heads = {"hm": 80, "wh": 2, "reg": 2}
m = create_model(arch, dict(heads), 256 if "dla" in arch else 64)
synth.fill_state_dict_(m, 317)
dev = torch.device("cuda:0")
m = m.to(dev).eval()
with torch.no_grad():
    m(synth.images(B, 512, 512, seed=0).to(dev))
plan = m.plan_for(B, 512, 512, dev)
for i, ((kind, act), meta) in enumerate(zip(plan.b.trace, plan.b.meta)):
    print("%3d %-9s -> (%d,%d,%d) pitch %d %s%s" % (i, kind, act.H, act.W, act.C, act.pitch, act.fmt, " nchw" if act.nchw else ""))
