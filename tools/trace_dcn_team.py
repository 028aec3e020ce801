"""Cycle stamps of a tile's life in the team form of the deformable kernel (cn_dcn3.hip probe build, key 9 bit 256): wave 0 of every
64th workgroup -- records (offset / mask loads + record arithmetic), first window, taps + window swaps, epilogue, store drain -- and
the wall-clock order in which the workgroups ended.  GPU box.   SHAPE=128,64,64,64 python tools/trace_dcn_team.py"""
import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act, exponent_for
dev = torch.device("cuda:0"); lib = native.lib()
ci, H, W, co = [int(v) for v in os.environ.get("SHAPE", "64,128,128,64").split(",")]
B = 32
lib.cn_set_tuning(23, 4); lib.cn_set_tuning(9, 256 | int(os.environ.get("DBG", "0")))
m = DCN(ci, co, (3, 3), 1, 1); synth.fill_state_dict_(m, 3)
xt = torch.randn((B, H, W, ci), device=dev).relu_()
pb = PlanBuilder(dev, B, H, W, exps={"x": exponent_for(float(xt.max())), "t1": exponent_for(8.0)})
x = Act(xt, B, H, W, ci, exp=pb._exp("x"), lid="x"); pb.dcn(x, m, relu=True)
for op in pb.ops: op()
for _ in range(3): pb.ops[-1]()
torch.cuda.synchronize()
out = np.zeros(64 * 8, np.uint64)
lib.cn_dcn_team_trace.argtypes = [ctypes.c_void_p]
assert lib.cn_dcn_team_trace(out.ctypes.data) == 0
tr = out.reshape(64, 8).astype(np.int64)
print("workgroup (every 64th): cycles  records | window wait | barrier | taps+swaps | epilogue | store drain | total   ; end wall (us after the first to end)")
t0 = tr[:, 7][tr[:, 7] > 0].min()
for i in range(64):
    if tr[i, 0] == 0: continue
    d = [int(tr[i, e] - tr[i, e - 1]) for e in range(1, 7)]
    print("  wg %4d" % (i * 64), " ".join("%7d" % v for v in d), " %7d" % int(tr[i, 6] - tr[i, 0]), "   %8.1f" % ((tr[i, 7] - t0) / 100.0))
lib.cn_set_tuning(9, 0); lib.cn_set_tuning(23, 0)
