"""Ablation of the LDS-halo kernel (cn_set_tuning key 9): 1 = no epilogue, 2 = no weight
staging after the first tap, 4 = no halo staging after the first chunk.  Results are wrong by
construction; only the timing is read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native
from centernet_amd.engine import PlanBuilder, Act
dev = torch.device("cuda:0"); lib = native.lib(); B = 32
SHAPES = [(64, 128, 128, 64), (64, 128, 128, 192), (128, 64, 64, 128), (256, 32, 32, 256), (512, 16, 16, 512)]
ONLY = [int(v) for v in os.environ.get("ONLY", "").split(",") if v]
MODES = [int(v) for v in os.environ.get("MODES", "0,1,2,4,7").split(",")]
if ONLY:
    SHAPES = [SHAPES[i] for i in ONLY]
for kv in [t for t in os.environ.get("TUNE", "").split(",") if t]:
    lib.cn_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
STAG = [int(v) for v in os.environ.get("STAG", "").split(",") if v]
print("%-22s" % "Cin,H,W,Cout", "  ".join("%-12s" % n for n in (["stag%d" % v for v in STAG] if STAG else ["dbg%d" % m for m in MODES])))
for ci, H, W, co in SHAPES:
    x = Act(torch.randn((B, H, W, ci), device=dev), B, H, W, ci)
    if os.environ.get("CN_F32S", "1") != "0":       # f32s activations, as inside the network
        pb0 = PlanBuilder(dev, B, H, W)
        x = pb0.packed(x)
        for op in pb0.ops: op()
        torch.cuda.synchronize()
    w = torch.randn((co, ci, 3, 3)) * 0.05
    row = []
    for dbg in (STAG if STAG else MODES):
        lib.cn_set_tuning(18 if STAG else 9, dbg)
        pb = PlanBuilder(dev, B, H, W)
        pb.conv(x, w, relu=True, stride=1, padding=1, out_plain=os.environ.get("OUT_PLAIN", "0") == "1")
        for _ in range(3): pb.ops[0]()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): pb.ops[0]()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        row.append("%.3f %5.1fTF" % (ms, pb.flops / ms / 1e9))
    print("%-22s" % str((ci, H, W, co)), "  ".join(row))
lib.cn_set_tuning(9, 0); lib.cn_set_tuning(18, 0)
