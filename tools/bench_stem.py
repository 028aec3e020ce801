"""The fused stem + max-pool launch of the ResNet backbones (B = 32, 512 x 512) under the probe switches of
cn_set_tuning key 43 (1 = no MFMAs, 2 = no window stores, 4 = no image loads, 8 = no pooling / stores, 16 = no
barriers).  GPU box.   DBG=0,1,2,4,8,16 python tools/bench_stem.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.engine import PlanBuilder
dev = torch.device("cuda:0"); lib = native.lib()
B, H, W, Cout = int(os.environ.get("B", "32")), 512, 512, 64
x = synth.images(B, H, W, 3)
w = torch.from_numpy(synth.normal((Cout, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
bn = torch.nn.BatchNorm2d(Cout).eval()
pb = PlanBuilder(dev, B, H, W, split=True)
y = pb.conv(pb.set_input(3), w, bn=bn, relu=True, stride=2, padding=3, pool=(3, 2, 1))
pb.input.t = x.to(dev)
assert len(pb.ops) == 1, len(pb.ops)
op = pb.ops[0]
KEY = int(os.environ.get("KEY", "43"))     # 43 = probe switches, 44 = start delay of the second resident workgroup
vals = [int(v) for v in os.environ.get("DBG", "0,1,2,4,8,16,3,7,15,31").split(",")]
times = {v: [] for v in vals}
for _ in range(int(os.environ.get("ROUNDS", "5"))):
    for v in vals:
        lib.cn_set_tuning(KEY, v)
        for _ in range(3): op()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): op()
        e.record(); torch.cuda.synchronize()
        times[v].append(s.elapsed_time(e) / 20)
lib.cn_set_tuning(KEY, 0)
for v in vals:
    t = sorted(times[v])[len(times[v]) // 2]
    print("key%d = %2d   %.3f ms   (%.1f TFLOP/s nominal)" % (KEY, v, t, 2 * B * 256 * 256 * 64 * 147 / t / 1e9))
