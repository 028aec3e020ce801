"""Cycle stamps of the wide deformable kernel (cn_dcn4.hip probe build, key 9 bit 512): where waves 0 and 4
of workgroup 0 spend a step.  GPU box.   SHAPE=256,32,32,256 python tools/trace_dcn_wide.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act, exponent_for
dev = torch.device("cuda:0"); lib = native.lib()
ci, H, W, co = [int(v) for v in os.environ.get("SHAPE", "256,32,32,256").split(",")]
B = 32
lib.cn_set_tuning(23, int(os.environ.get("FORM", "6"))); lib.cn_set_tuning(9, 512 | int(os.environ.get("DBG", "0")))
m = DCN(ci, co, (3, 3), 1, 1); synth.fill_state_dict_(m, 3)
xt = torch.randn((B, H, W, ci), device=dev).relu_()
pb = PlanBuilder(dev, B, H, W, exps={"x": exponent_for(float(xt.max())), "t1": exponent_for(8.0)})
x = Act(xt, B, H, W, ci, exp=pb._exp("x"), lid="x"); pb.dcn(x, m, relu=True)
for op in pb.ops: op()
for _ in range(3): pb.ops[-1]()
torch.cuda.synchronize()
out = np.zeros(2 * 64 * 8, np.uint64)
lib.cn_dcn_wide_trace.argtypes = [ctypes.c_void_p]
assert lib.cn_dcn_wide_trace(out.ctypes.data) == 0
tr = out.reshape(2, 64, 8).astype(np.int64)
names = ["top", "vmcnt", "barrier", "blend1", "dma", "request", "mfma", "blend0"]
for k in range(2):
    print("team", k, "(wave %d): per step deltas" % (4 * k), names[1:], "| step to step")
    for s in range(1, 28):
        d = [int(tr[k, s, e] - tr[k, s, e - 1]) if tr[k, s, e] and tr[k, s, e - 1] else 0 for e in range(1, 8)]
        print("  step %2d" % s, " ".join("%6d" % v for v in d), " | %6d" % int(tr[k, s, 0] - tr[k, s - 1, 0]), " offset vs team0 %6d" % int(tr[k, s, 0] - tr[0, s, 0]))
lib.cn_set_tuning(9, 0); lib.cn_set_tuning(23, 0)
