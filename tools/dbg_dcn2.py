"""Debug: one tiny launch of the LDS-window deformable kernel with the buffer addresses printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCNv2
from centernet_amd.engine import PlanBuilder, Act, exponent_for
dev = torch.device("cuda:0")
lib = native.lib()
B, C, H, W, Co = [int(v) for v in os.environ.get("SHAPE", "1,32,8,8,64").split(",")]
rs = np.random.RandomState(0)
x = rs.standard_normal((B, C, H, W)).astype(np.float32)
off = (rs.standard_normal((B, 18, H, W)) * float(os.environ.get("STD", "0.5"))).astype(np.float32)
mask = rs.uniform(0, 1, (B, 9, H, W)).astype(np.float32)
w = (rs.standard_normal((Co, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
b = rs.standard_normal(Co).astype(np.float32)
m = DCNv2(C, Co, (3, 3), 1, 1)
with torch.no_grad():
    m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
m.conv_offset_mask = None
om = np.zeros((B, H, W, 32), np.float32)
om[..., :18] = off.transpose(0, 2, 3, 1); om[..., 18:27] = mask.transpose(0, 2, 3, 1)
lib.cn_set_tuning(23, int(os.environ.get("FORM", "2"))); lib.cn_set_tuning(9, int(os.environ.get("DBG", "0")))
pb = PlanBuilder(dev, B, H, W, split=True, exps={"x": exponent_for(float(np.abs(x).max())), "t1": exponent_for(16.0)})
xa = Act(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev), B, H, W, C, exp=pb._exp("x"), lid="x")
oma = Act(torch.from_numpy(om).to(dev), B, H, W, 27, pitch=32)
y = pb.dcn(xa, m, om=oma, mask_sigmoid=False, out_plain=os.environ.get("PLAIN", "0") == "1")
print("x %x +%d | om %x +%d | y %x +%d | range %x | ws %s" % (
    xa.t.data_ptr(), xa.t.numel() * 4, oma.t.data_ptr(), oma.t.numel() * 4, y.t.data_ptr(), y.t.numel() * 4,
    pb.range.data_ptr() if pb.range is not None else 0, pb.ws_bytes), flush=True)
for t in pb.keep:
    if torch.is_tensor(t):
        print("keep %x +%d" % (t.data_ptr(), t.numel() * t.element_size()), flush=True)
for op in pb.ops:
    op()
torch.cuda.synchronize()
from oracle import cref
want = cref.dcn_v2_forward(x, off, mask, w, b)
got = y.to_float().permute(0, 3, 1, 2).cpu().numpy()
print("max err", float(np.abs(got - want).max()), "ref max", float(np.abs(want).max()), flush=True)
