"""Identity-weight probe of the wave-specialised window kernel (cn_dcn2.hip dcn_win_kernel, form 3):
zero offsets, unit masks, centre-tap identity weights, so y == x; cn_set_tuning key 9 switches replace
stages by constants (32 stores, 64 MFMA operands, 128 window reads, 256 the sampling record itself).
This is how the record fields were found to read element 0 (bit_cast on a vector element).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import native
from centernet_amd.dcn_v2 import DCNv2
from centernet_amd.engine import PlanBuilder, Act, exponent_for
dev = torch.device("cuda:0"); lib = native.lib()
B, C, H, W, Co = 1, 32, 8, 8, 64
rs = np.random.RandomState(0)
def run(x, off, mask, w, b, dbg):
    m = DCNv2(C, Co, (3, 3), 1, 1)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    m.conv_offset_mask = None
    om = np.zeros((B, H, W, 32), np.float32)
    om[..., :18] = off.transpose(0, 2, 3, 1); om[..., 18:27] = mask.transpose(0, 2, 3, 1)
    lib.cn_set_tuning(23, 3); lib.cn_set_tuning(9, 0)
    pb = PlanBuilder(dev, B, H, W, split=True, exps={"x": 0, "t1": 0})
    xa = Act(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev), B, H, W, C, exp=0, lid="x")
    oma = Act(torch.from_numpy(om).to(dev), B, H, W, 27, pitch=32)
    y = pb.dcn(xa, m, om=oma, mask_sigmoid=False, out_plain=True)
    y.t.fill_(-5.0)
    # key 9 accepts 0..31 only through cn_set_tuning: widen by calling twice is not possible -> use env-free path
    rc = lib.cn_set_tuning(9, dbg)
    for op in pb.ops: op()
    torch.cuda.synchronize()
    lib.cn_set_tuning(9, 0); lib.cn_set_tuning(23, 0)
    return rc, y.to_float().permute(0, 3, 1, 2).cpu().numpy()
x = (np.arange(B * C * H * W, dtype=np.float32).reshape(B, C, H, W) % 97) / 97.0 + 0.5
off = np.zeros((B, 18, H, W), np.float32); mask = np.ones((B, 9, H, W), np.float32)
w = np.zeros((Co, C, 3, 3), np.float32)
for o in range(Co): w[o, o % C, 1, 1] = 1.0
b0 = np.zeros(Co, np.float32)
for dbg in (0, 32, 64, 128, 256):
    rc, y = run(x, off, mask, w, b0, dbg)
    print("dbg", dbg, "rc", rc, "y[0,:4,0,:4]=", np.round(y[0, :4, 0, :4], 3).tolist(), "y[0,33,3,:4]", np.round(y[0, 33, 3, :4], 3).tolist(), "x[0,:4,0,:4]", np.round(x[0, :4, 0, :4], 3).tolist(), flush=True)
