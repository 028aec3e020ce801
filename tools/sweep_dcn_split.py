"""Deformable layers: K-split workgroup target (cn_set_tuning key 34) x 64-wide N tiles (key 35), real
offset maps, B = 32.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act, exponent_for

dev = torch.device("cuda:0")
lib = native.lib()
COMBOS = [(256, 0), (512, 0), (1024, 0), (256, 4096), (512, 4096), (1024, 4096)]
SHAPES = [(512, 16, 16, 256), (256, 32, 32, 128), (128, 64, 64, 64), (256, 32, 32, 256), (128, 64, 64, 128)]
B = 32
print("B=%d %-22s" % (B, "Cin,H,W,Cout"), "  ".join("wgs%4d/bn64<%-5d" % c for c in COMBOS))
for ci, H, W, co in SHAPES:
    m = DCN(ci, co, (3, 3), 1, 1)
    synth.fill_state_dict_(m, 3)
    xt = torch.randn((B, H, W, ci), device=dev).relu_()
    ops = []
    for wg, bn in COMBOS:
        lib.cn_set_tuning(34, wg); lib.cn_set_tuning(35, bn)
        pb = PlanBuilder(dev, B, H, W, exps={"x": exponent_for(float(xt.max())), "t1": exponent_for(8.0)})
        x = Act(xt, B, H, W, ci, exp=pb._exp("x"), lid="x")
        pb.dcn(x, m, relu=True)
        for op in pb.ops:
            op()
        ops.append((pb.ops[-1], pb.meta[-1]["flops"], pb))
    best = [1e9] * len(COMBOS)
    for rnd in range(5):                     # interleaved rounds: the box's clock drifts
        for i, ((wg, bn), (op, fl, _)) in enumerate(zip(COMBOS, ops)):
            lib.cn_set_tuning(34, wg); lib.cn_set_tuning(35, bn)
            op(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): op()
            e.record(); torch.cuda.synchronize()
            best[i] = min(best[i], s.elapsed_time(e) / 10)
    print("     %-22s" % str((ci, H, W, co)), "  ".join("%.3fms %5.1fTF   " % (b, ops[0][1] / b / 1e9) for b in best))
lib.cn_set_tuning(34, 256); lib.cn_set_tuning(35, 0)
