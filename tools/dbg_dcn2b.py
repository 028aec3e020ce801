"""Debug: structured cases for the LDS-window deformable kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import native
from centernet_amd.dcn_v2 import DCNv2
from centernet_amd.engine import PlanBuilder, Act, exponent_for
from oracle import cref
dev = torch.device("cuda:0")
lib = native.lib()
B, C, H, W, Co = [int(v) for v in os.environ.get("SHAPE", "1,32,8,8,64").split(",")]
rs = np.random.RandomState(0)
def run(x, off, mask, w, b, form=2):
    m = DCNv2(C, Co, (3, 3), 1, 1)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    m.conv_offset_mask = None
    om = np.zeros((B, H, W, 32), np.float32)
    om[..., :18] = off.transpose(0, 2, 3, 1); om[..., 18:27] = mask.transpose(0, 2, 3, 1)
    lib.cn_set_tuning(23, form)
    pb = PlanBuilder(dev, B, H, W, split=True, exps={"x": exponent_for(float(np.abs(x).max())), "t1": exponent_for(64.0)})
    xa = Act(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev), B, H, W, C, exp=pb._exp("x"), lid="x")
    oma = Act(torch.from_numpy(om).to(dev), B, H, W, 27, pitch=32)
    y = pb.dcn(xa, m, om=oma, mask_sigmoid=False, out_plain=True)
    for op in pb.ops: op()
    torch.cuda.synchronize()
    return y.to_float().permute(0, 3, 1, 2).cpu().numpy()
x = rs.standard_normal((B, C, H, W)).astype(np.float32)
zero_off = np.zeros((B, 18, H, W), np.float32)
one_mask = np.ones((B, 9, H, W), np.float32)
b0 = np.zeros(Co, np.float32)
for tap in range(9):
    w = np.zeros((Co, C, 3, 3), np.float32)
    for o in range(Co): w[o, o % C, tap // 3, tap % 3] = 1.0 + 0.01 * o
    got = run(x, zero_off, one_mask, w, b0)
    want = cref.dcn_v2_forward(x, zero_off, one_mask, w, b0)
    err = np.abs(got - want)
    print("tap", tap, "max err %.3g" % err.max(), "bad pixels rows", np.unique(np.argwhere(err.max(axis=(0, 1)) > 1e-3)[:, 0]).tolist(),
          "cols", np.unique(np.argwhere(err.max(axis=(0, 1)) > 1e-3)[:, 1]).tolist(), "bad channels", np.argwhere(err.max(axis=(0, 2, 3)) > 1e-3)[:, 0].tolist()[:12], flush=True)
w = (rs.standard_normal((Co, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
for name, off in (("zero-off", zero_off), ("0.3-off", zero_off + 0.3), ("rand0.5", (rs.standard_normal((B, 18, H, W)) * 0.5).astype(np.float32))):
    got = run(x, off, one_mask, w, b0); want = cref.dcn_v2_forward(x, off, one_mask, w, b0)
    g1 = run(x, off, one_mask, w, b0, form=1)
    print(name, "window err %.3g  gather err %.3g" % (np.abs(got - want).max(), np.abs(g1 - want).max()), flush=True)
