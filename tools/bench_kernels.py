"""Micro-benchmark of single launches (conv shapes of resdcn_18 at B=32) under the
tuning knobs of cn_set_tuning.  GPU box only.  Usage: python tools/bench_kernels.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from centernet_amd import native, synth
from centernet_amd.engine import PlanBuilder, Act

dev = torch.device("cuda:0")
lib = native.lib()
B = int(os.environ.get("B", 32))
SHAPES = [  # Cin, H, W, Cout, k, stride
    (64, 128, 128, 64, 3, 1), (64, 128, 128, 128, 3, 2), (128, 64, 64, 128, 3, 1),
    (128, 64, 64, 256, 3, 2), (256, 32, 32, 256, 3, 1), (512, 16, 16, 512, 3, 1),
    (64, 128, 128, 192, 3, 1), (512, 16, 16, 27, 3, 1), (128, 64, 64, 27, 3, 1),
    (64, 128, 128, 128, 1, 2), (256, 32, 32, 512, 1, 2),
]


def time_ops(pb, iters=20):
    for _ in range(3):
        for op in pb.ops:
            op()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        for op in pb.ops:
            op()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


knobs = [(10, 1), (10, 0)]
if os.environ.get('ONLY'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['ONLY'].split(',')]
if os.environ.get('NBUF'):
    knobs = [(1, int(os.environ['NBUF']))]
print("%-34s" % "shape (Cin,H,W,Cout,k,s)", *["key%d=%d" % kv for kv in knobs])
for (ci, H, W, co, k, s) in SHAPES:
    x = Act(torch.randn((B, H, W, ci), device=dev), B, H, W, ci)
    w = torch.randn((co, ci, k, k)) * 0.05
    row = []
    for key, val in knobs:
        assert lib.cn_set_tuning(key, val) == 0
        pb = PlanBuilder(dev, B, H, W)
        pb.conv(x, w, relu=True, stride=s, padding=k // 2)
        ms = time_ops(pb)
        row.append("%6.3f ms %6.1f TF" % (ms, pb.flops / ms / 1e9))
    print("%-34s" % str((ci, H, W, co, k, s)), *row)
lib.cn_set_tuning(1, 0); lib.cn_set_tuning(4, 0); lib.cn_set_tuning(9, 0); lib.cn_set_tuning(10, 0)
