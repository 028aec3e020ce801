"""DCN layer micro-benchmark under the tuning knobs (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act

dev = torch.device("cuda:0")
lib = native.lib()
B = int(os.environ.get("B", 32))
SHAPES = [(512, 16, 16, 256), (256, 32, 32, 128), (128, 64, 64, 64), (64, 128, 128, 64), (256, 32, 32, 64)]
print("%-24s" % "Cin,H,W,Cout", "gather-kernel(tile64,nbuf1)   window-kernel")
for ci, H, W, co in SHAPES:
    m = DCN(ci, co, (3, 3), 1, 1)
    synth.fill_state_dict_(m, 3)
    x = Act(torch.randn((B, H, W, ci), device=dev), B, H, W, ci)
    row = []
    for gather in (0, 1):
        for nbuf in (1,):
            lib.cn_set_tuning(11, gather); lib.cn_set_tuning(3, 64); lib.cn_set_tuning(1, nbuf)
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
    print("%-24s" % str((ci, H, W, co)), "  ".join(row))
lib.cn_set_tuning(3, 0); lib.cn_set_tuning(1, 0); lib.cn_set_tuning(11, 0)
