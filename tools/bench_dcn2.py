"""Deformable layer micro-benchmark with REAL offset maps (the offset convolution of the synthetic
DCN module runs first) under cn_set_tuning knobs.  GPU box only.
    KNOB=22 VALUES=0,1 python tools/bench_dcn2.py        # default: tile shape A/B
Columns: one per knob value; rows: the resdcn_18 / dla_34 layer shapes at B = 32 (and B = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native, synth
from centernet_amd.dcn_v2 import DCN
from centernet_amd.engine import PlanBuilder, Act, exponent_for

dev = torch.device("cuda:0")
lib = native.lib()
KNOB = int(os.environ.get("KNOB", "22"))
VALUES = [int(v) for v in os.environ.get("VALUES", "0,1").split(",")]
# DBG=0,8,128: columns are probe builds of a window kernel (key 23 = FORM, default 2; key 9 = value)
DBG = [int(v) for v in os.environ["DBG"].split(",")] if os.environ.get("DBG") else None
ZERO_OFF = os.environ.get("ZERO_OFF") == "1"     # zero offsets (no LDS bank conflicts, nothing beyond the window)
# COLD=1: every timed launch runs behind a 1 GiB copy (weights and input gone from L2 / the Infinity Cache, as inside a
# network step where 4 GB pass between two uses of a layer's weights) and is timed alone
COLD = os.environ.get("COLD") == "1"
if COLD:
    _ca = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    _cb = torch.empty(256 << 20, dtype=torch.float32, device=dev)
if DBG:
    KNOB, VALUES = 9, DBG
    lib.cn_set_tuning(23, int(os.environ.get("FORM", "2")))
SHAPES = [(512, 16, 16, 256), (256, 32, 32, 128), (128, 64, 64, 64), (64, 128, 128, 64), (256, 32, 32, 64),
          (128, 64, 64, 128), (256, 32, 32, 256)]
if os.environ.get("ONLY"):     # ONLY=2,3: a subset of the shapes (counter passes)
    SHAPES = [SHAPES[int(i)] for i in os.environ["ONLY"].split(",")]
for B in [int(b) for b in os.environ.get("B", "32").split(",")]:
    print("B=%d  %-22s" % (B, "Cin,H,W,Cout"), "   ".join("key%d=%-10d" % (KNOB, v) for v in VALUES))
    for ci, H, W, co in SHAPES:
        m = DCN(ci, co, (3, 3), 1, 1)
        synth.fill_state_dict_(m, 3)
        if ZERO_OFF:
            with torch.no_grad():
                m.conv_offset_mask.weight.zero_(); m.conv_offset_mask.bias.zero_()
        xt = torch.randn((B, H, W, ci), device=dev).relu_()
        row = []
        plans = []
        for v in VALUES:
            lib.cn_set_tuning(KNOB, v)
            pb = PlanBuilder(dev, B, H, W, exps={"x": exponent_for(float(xt.max())), "t1": exponent_for(8.0)})
            x = Act(xt, B, H, W, ci, exp=pb._exp("x"), lid="x")
            pb.dcn(x, m, relu=True)
            for op in pb.ops:        # offset conv -> real offsets (sigma ~ 1.4 px), then the DCN
                op()
            plans.append(pb)
        # interleaved rounds (ROUNDS, default 5), median per column: boxes drift by several per cent within seconds
        times = [[] for _ in VALUES]
        for _ in range(int(os.environ.get("ROUNDS", "5"))):
            for k, v in enumerate(VALUES):
                lib.cn_set_tuning(KNOB, v)
                op = plans[k].ops[-1]
                for _ in range(3): op()
                torch.cuda.synchronize()
                if COLD:
                    tot = 0.0
                    for _ in range(8):
                        _cb.copy_(_ca)
                        plans[k].ops[-2]()       # the offset conv just in front, as in the network (it reads the input)
                        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s.record(); op(); e.record(); torch.cuda.synchronize()
                        tot += s.elapsed_time(e)
                    times[k].append(tot / 8)
                    continue
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20): op()
                e.record(); torch.cuda.synchronize()
                times[k].append(s.elapsed_time(e) / 20)
        for k in range(len(VALUES)):
            ms = sorted(times[k])[len(times[k]) // 2]
            row.append("%.3fms %5.1fTF" % (ms, plans[k].meta[-1]["flops"] / ms / 1e9))
        print("      %-22s" % str((ci, H, W, co)), "   ".join("%-16s" % r for r in row))
lib.cn_set_tuning(KNOB, 1 if KNOB in (22, 23) else 0)
lib.cn_set_tuning(23, 0)
