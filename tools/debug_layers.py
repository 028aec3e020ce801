"""Layer-by-layer divergence of the HIP plan vs the CPU oracle (debugging aid, GPU box)."""
import sys, os
os.environ.setdefault("CN_FUSE_STEM_POOL", "0")   # the oracle traces the stem and the pool separately
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from centernet_amd import synth
from centernet_amd.model import create_model
from oracle import net_oracle

arch = sys.argv[1] if len(sys.argv) > 1 else "resdcn_18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
heads = {"hm": 80, "wh": 2, "reg": 2}
m = create_model(arch, dict(heads), 64)
synth.fill_state_dict_(m, 317)
dev = torch.device("cuda:0")
m = m.to(dev).eval()
x = synth.images(B, R, R, seed=0)
with torch.no_grad():
    out = m(x.to(dev))[-1]
torch.cuda.synchronize()
plan = m.plan_for(B, R, R, dev)
net_oracle.TRACE = []
ref = net_oracle.forward(arch, m.state_dict(), x, list(heads))
tr = net_oracle.TRACE
# GPU trace: drop the offset-mask convs (27 channels, pitch 32), and group 4 deconv launches
gtr = []
for kind, act in plan.b.trace:
    if kind == "convert" or (act.C == 27 and act.pitch == 32):
        continue
    if act.nchw:
        continue
    if gtr and gtr[-1][1] is act:
        continue
    gtr.append((kind, act))
print(len(tr), len(gtr))
for (name, r), (kind, act) in zip(tr, gtr):
    g = act.to_float().permute(0, 3, 1, 2).cpu()
    if g.shape != r.shape:
        print("SHAPE", name, kind, tuple(g.shape), tuple(r.shape)); continue
    d = (g - r).abs()
    print("%-22s %-8s shape %-20s |ref| max %9.3g rms %9.3g   err max %9.3g rms %9.3g  rel(rms) %8.2e" % (
        name, kind, tuple(r.shape), r.abs().max(), r.pow(2).mean().sqrt(), d.max(), d.pow(2).mean().sqrt(),
        d.pow(2).mean().sqrt() / (r.pow(2).mean().sqrt() + 1e-30)))
for h in heads:
    g = out[h].cpu(); r = ref[h]
    d = (g - r).abs()
    print("%-22s head     |ref| max %9.3g rms %9.3g  err max %9.3g rms %9.3g" % (h, r.abs().max(), r.pow(2).mean().sqrt(), d.max(), d.pow(2).mean().sqrt()))
