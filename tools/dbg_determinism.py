"""Run-to-run determinism of the f32s kernels under co-resident workgroups (GPU box)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from centernet_amd import synth, native
from centernet_amd.engine import PlanBuilder, Act
from centernet_amd.dcn_v2 import DCN
dev = torch.device("cuda:0")
lib = native.lib()
B = 32

def rep(name, pb, out, n=30):
    outs = None; bad = 0
    for t in range(n):
        for op in pb.ops: op()
        torch.cuda.synchronize()
        cur = out.t.clone()
        if outs is None: outs = cur
        elif not torch.equal(cur.view(torch.int32), outs.view(torch.int32)): bad += 1
    print("%-40s non-identical repeats: %d of %d" % (name, bad, n - 1))

for split in (True, False):
    tag = "f32s " if split else "fp32 "
    x = Act(torch.randn((B, 128, 128, 64), device=dev).relu_(), B, 128, 128, 64)
    w = torch.randn((64, 64, 3, 3)) * 0.05
    pb = PlanBuilder(dev, B, 128, 128, split=split); y = pb.conv(x, w, relu=True, stride=1, padding=1); rep(tag + "halo 64->64@128", pb, y)
    w = torch.randn((128, 64, 3, 3)) * 0.05
    pb = PlanBuilder(dev, B, 128, 128, split=split); y = pb.conv(x, w, relu=True, stride=2, padding=1); rep(tag + "igemm 3x3/s2 64->128", pb, y)
    w = torch.randn((128, 64, 1, 1)) * 0.05
    pb = PlanBuilder(dev, B, 128, 128, split=split); y = pb.conv(x, w, stride=2, padding=0); rep(tag + "igemm 1x1/s2 64->128", pb, y)
    for (C, H, Co) in [(128, 64, 64), (256, 32, 128), (512, 16, 256)]:
        m = DCN(C, Co, (3, 3), 1, 1); synth.fill_state_dict_(m, 5)
        xa = Act(torch.randn((B, H, H, C), device=dev).relu_(), B, H, H, C)
        for ks in (0, 1, 3, 9):
            lib.cn_set_tuning(13, ks)
            pb = PlanBuilder(dev, B, H, H, split=split); y = pb.dcn(xa, m, out_plain=True)
            rep(tag + "dcn %d->%d@%d tap-split knob %d" % (C, Co, H, ks), pb, y, 20)
        lib.cn_set_tuning(13, 0)
