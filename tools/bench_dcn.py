"""DCN layer micro-benchmark under the tuning knobs (GPU box only).
Columns: tap split off / auto (key 13), LDS tile buffers 1 / 2 (key 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act

dev = torch.device("cuda:0")
lib = native.lib()
SHAPES = [(512, 16, 16, 256), (256, 32, 32, 128), (128, 64, 64, 64), (64, 128, 128, 64), (256, 32, 32, 64)]
CFGS = [("split1,nbuf1", 1, 1), ("auto,nbuf1", 0, 1), ("split3,nbuf1", 3, 1), ("auto,nbuf2", 0, 2)]
if os.environ.get("SWZ"):
    CFGS = [("auto,no-xcd", 0, 1), ("auto+xcd", 0, 1)]
for B in [int(b) for b in os.environ.get("B", "32,1").split(",")]:
    print("B=%d  %-22s" % (B, "Cin,H,W,Cout"), "   ".join("%-16s" % c[0] for c in CFGS))
    for ci, H, W, co in SHAPES:
        m = DCN(ci, co, (3, 3), 1, 1)
        synth.fill_state_dict_(m, 3)
        x = Act(torch.randn((B, H, W, ci), device=dev), B, H, W, ci)
        row = []
        for _, split, nbuf in CFGS:
            lib.cn_set_tuning(13, split); lib.cn_set_tuning(1, nbuf)
            lib.cn_set_tuning(7, 0 if _.endswith('+xcd') else 2)
            pb = PlanBuilder(dev, B, H, W)
            pb.dcn(x, m, relu=True)
            op = pb.ops[-1]          # the deformable launch (ops[0] is the offset conv)
            for _ in range(3): op()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): op()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 20
            row.append("%.3fms %5.1fTF" % (ms, pb.meta[-1]["flops"] / ms / 1e9))
        print("      %-22s" % str((ci, H, W, co)), "   ".join("%-16s" % r for r in row))
lib.cn_set_tuning(13, 0); lib.cn_set_tuning(1, 0); lib.cn_set_tuning(7, 0)
