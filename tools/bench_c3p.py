"""The persistent loader / consumer 3x3 kernel (cn_conv3x3p.hip, cn_set_tuning key 28) against the
one-tile-per-workgroup halo kernel: correctness vs torch fp64 on odd and edge shapes, then time at
the resdcn_18 / dla_34 trunk shapes.  GPU box only.

usage: python tools/bench_c3p.py            # correctness + timing
       QUICK=1 python tools/bench_c3p.py    # correctness only
       STAG=0,8,16 python tools/bench_c3p.py   # also sweep the start delay of the second workgroup (key 29)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from centernet_amd import native
from centernet_amd.engine import prescale_rows
from centernet_amd.native import (ConvDesc, F32sCtl, LAYOUT_NHWC, DTYPE_F32S, CONV_Y_PLAIN, CONV_R_PLAIN)

dev = torch.device("cuda:0")
lib = native.lib()
st = native.stream_ptr


def to_f32s(t):
    B, H, W, C = t.shape
    pitch = (C + 31) // 32 * 32
    out = torch.zeros((B, H, W, pitch), device=dev)
    native.check(lib.cn_f32_to_f32s(native.ptr(t.contiguous()), native.ptr(out), B * H * W, C, C, pitch, st()), "cvt")
    return out


def from_f32s(t, C):
    B, H, W, pitch = t.shape
    out = torch.empty((B, H, W, pitch), device=dev)
    native.check(lib.cn_f32s_to_f32(native.ptr(t), native.ptr(out), B * H * W, C, pitch, pitch, st()), "cvt")
    return out[..., :C]


class Case:
    def __init__(self, B, ci, H, W, co, res=0, out_plain=False, relu=True, seed=0):
        self.B, self.ci, self.H, self.W, self.co = B, ci, H, W, co
        self.res, self.out_plain, self.relu = res, out_plain, relu
        g = torch.Generator().manual_seed(seed + ci + co + H)
        self.x = (torch.randn((B, H, W, ci), generator=g) * 8).to(dev)
        self.w = torch.randn((co, ci, 3, 3), generator=g) * (2.0 / (ci * 9)) ** 0.5
        self.scale = (torch.rand(co, generator=g) + 0.5)
        self.shift = torch.randn(co, generator=g)
        self.r = (torch.randn((B, H, W, co), generator=g) * 8).to(dev) if res else None
        ws, factor = prescale_rows(self.w)
        n = lib.cn_packed_conv_weight_elems(co, ci, 3, 3, DTYPE_F32S)
        self.wp = torch.empty(n, device=dev)
        native.check(lib.cn_pack_conv_weight(native.ptr(ws.to(dev).contiguous()), native.ptr(self.wp), co, ci, 3, 3,
                                             DTYPE_F32S, st()), "pack")
        self.sc = (self.scale * factor).to(dev).contiguous()
        self.sh = self.shift.to(dev).contiguous()
        self.xs = to_f32s(self.x)
        self.pitch_o = (co + 31) // 32 * 32
        if res == 1:
            self.rs = to_f32s(self.r)
        elif res == 2:
            self.rs = torch.zeros((B, H, W, self.pitch_o), device=dev)
            self.rs[..., :co] = self.r
        else:
            self.rs = None
        self.y = torch.zeros((B, H, W, self.pitch_o), device=dev)
        flags = (CONV_Y_PLAIN if out_plain else 0) | (CONV_R_PLAIN if res == 2 else 0)
        self.d = ConvDesc(B=B, H=H, W=W, Cin=ci, Ho=H, Wo=W, Cout=co, KH=3, KW=3, stride=1, pad_h=1, pad_w=1,
                          dil=1, in_layout=LAYOUT_NHWC, in_pitch=self.xs.shape[-1], out_layout=LAYOUT_NHWC,
                          out_pitch=self.pitch_o, OH=H, OW=W, oy_mul=1, oy_add=0, ox_mul=1, ox_add=0,
                          relu=int(relu), dtype=DTYPE_F32S, flags=flags)
        self.range = torch.zeros(2 * 64 * 16, device=dev, dtype=torch.int32)
        self.d.ctl = F32sCtl(1.0, 1.0, self.range.data_ptr())

    def launch(self):
        rc = lib.cn_conv2d(ctypes.byref(self.d), native.ptr(self.xs), native.ptr(self.wp), native.ptr(self.sc),
                           native.ptr(self.sh), native.ptr(self.rs) if self.rs is not None else None,
                           native.ptr(self.y), None, 0, st())
        assert rc == 0, rc

    def result(self):
        return self.y[..., :self.co].clone() if self.out_plain else from_f32s(self.y, self.co)

    def reference(self, nb):
        x = self.x[:nb].permute(0, 3, 1, 2).double().cpu()
        ref = torch.nn.functional.conv2d(x, self.w.double(), padding=1).permute(0, 2, 3, 1)
        ref = ref * self.scale.double() + self.shift.double()
        if self.r is not None:
            ref = ref + self.r[:nb].double().cpu()
        return ref.relu() if self.relu else ref

    def time(self, iters=30):
        for _ in range(3):
            self.launch()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            self.launch()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters


def check(case, nb=None, label=""):
    nb = case.B if nb is None else min(nb, case.B)
    ref = case.reference(nb)
    rms = float(ref.pow(2).mean().sqrt())
    out = {}
    for key in (0, 2):
        lib.cn_set_tuning(28, key)
        case.y.zero_()
        case.range.zero_()
        case.launch()
        torch.cuda.synchronize()
        got = case.result()
        out[key] = (float((got[:nb].double().cpu() - ref).abs().max()) / rms, got,
                    float(case.range.view(torch.float32).max()))
    d01 = float((out[0][1] - out[2][1]).abs().max()) / rms
    # everything beyond the checked images must agree between the two kernels
    ok = out[2][0] < max(6e-6, 1.5 * out[0][0]) and d01 < max(6e-6, 1.5 * out[0][0])
    # pad channels of an f32s output must stay zero
    print("%-44s halo %.2e  persist %.2e  |halo-persist| %.2e  range %.4g / %.4g  %s" % (
        label, out[0][0], out[2][0], d01, out[0][2], out[2][2], "ok" if ok else "FAIL"))
    return ok


allok = True
lib.cn_set_tuning(30, int(os.environ.get("CHECK_KNOBS", "0")))     # key 30 for the correctness pass
print("== correctness (error relative to the rms of the fp64 reference)")
CASES = [
    ("B2 64->64 @32x32", dict(B=2, ci=64, H=32, W=32, co=64)),
    ("B1 64->64 @20x24 edge tiles", dict(B=1, ci=64, H=20, W=24, co=64)),
    ("B3 32->64 @17x40 one chunk, res f32s", dict(B=3, ci=32, H=17, W=40, co=64, res=1)),
    ("B2 96->96 @24x16 3 chunks, half block", dict(B=2, ci=96, H=24, W=16, co=96, res=1)),
    ("B2 128->128 @16x32 res plain", dict(B=2, ci=128, H=16, W=32, co=128, res=2)),
    ("B2 64->128 @16x16 out plain, no relu", dict(B=2, ci=64, H=16, W=16, co=128, out_plain=True, relu=False)),
    ("B2 80->64 @16x16 Cin pad, res f32s, out plain", dict(B=2, ci=80, H=16, W=16, co=64, res=1, out_plain=True)),
    ("B9 64->64 @8x16 many items per workgroup", dict(B=9, ci=64, H=64, W=64, co=64, res=1)),
    ("B32 64->64 @128 res f32s", dict(B=32, ci=64, H=128, W=128, co=64, res=1)),
    ("B32 512->512 @16", dict(B=32, ci=512, H=16, W=16, co=512, res=1)),
]
for label, kw in ([] if os.environ.get("SKIPCHECK") else CASES):
    c = Case(**kw)
    allok &= check(c, nb=2, label=label)
    # run-to-run bit equality of the persistent kernel
    lib.cn_set_tuning(28, 2)
    c.launch(); a = c.y.clone(); c.launch(); torch.cuda.synchronize()
    if not torch.equal(a, c.y):
        print("   NOT deterministic run to run")
        allok = False
    del c
print("ALL OK" if allok else "SOME FAILED")

if os.environ.get("QUICK"):
    sys.exit(0 if allok else 1)

# the pipelined fragment schedule (key 30 bit 2) sums in the order of the unpipelined one: bit-identical
# outputs, launch after launch (EQ = launches per shape; a fragment overwritten under a queued MFMA --
# the operand hazard of DESIGN 3.0 -- shows up here as a rare differing tile)
EQ = int(os.environ.get("EQ", "0"))
if EQ:
    print("== pipelined vs unpipelined schedule, bit equality over %d launches per shape" % EQ)
    lib.cn_set_tuning(28, 1)
    for (ci, H, W, co, res) in [(64, 128, 128, 64, 1), (128, 64, 64, 128, 0), (256, 32, 32, 256, 1), (512, 16, 16, 512, 0)]:
        c = Case(32, ci, H, W, co, res=res)
        lib.cn_set_tuning(30, 0)
        c.launch()
        base = c.y.clone()
        lib.cn_set_tuning(30, 2)
        bad = 0
        for _ in range(EQ):
            c.y.zero_()
            c.launch()
            bad += int(not torch.equal(c.y, base))
        print("%-24s res %d: %d of %d launches differ" % (str((ci, H, W, co)), res, bad, EQ))
        allok &= bad == 0
        del c
    lib.cn_set_tuning(30, 0)
    print("ALL OK" if allok else "SOME FAILED")

print("== timing, B = 32 (ms per launch, effective TFLOP/s)")
SHAPES = [(64, 128, 128, 64), (128, 64, 64, 128), (256, 32, 32, 256), (512, 16, 16, 512)]
stags = [int(v) for v in os.environ.get("STAG", "64").split(",")]
knobs = [int(v) for v in os.environ.get("KNOBS", "0").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "5"))
for (ci, H, W, co) in SHAPES:
    for res in (0, 1):
        c = Case(32, ci, H, W, co, res=res)
        fl = 2.0 * 32 * H * W * co * ci * 9
        # configurations are timed in interleaved rounds and reported by their median: the clock the part
        # holds drifts by several per cent over a second, so back-to-back blocks favour whoever runs last
        cfgs = [("halo", 0, 64, 0)] + [("persist(stag %d knobs %d)" % (sg, kn), 1, sg, kn) for sg in stags for kn in knobs]
        times = {name: [] for name, *_ in cfgs}
        for _ in range(ROUNDS):
            for name, k28, sg, kn in cfgs:
                lib.cn_set_tuning(28, k28)
                lib.cn_set_tuning(29, sg)
                lib.cn_set_tuning(30, kn)
                times[name].append(c.time(10))
        row = []
        for name, *_ in cfgs:
            ms = sorted(times[name])[len(times[name]) // 2]
            row.append("%s %.4f ms %6.1f TF" % (name, ms, fl / ms / 1e9))
        print("%-22s res %d | %s" % (str((ci, H, W, co)), res, " | ".join(row)))
        del c
lib.cn_set_tuning(28, 1)
lib.cn_set_tuning(29, 64)
lib.cn_set_tuning(30, 0)

# ---- where a launch spends its time: ablation switches and in-kernel cycle counters of the
# instrumented instantiation (cn_conv3x3p_probe)
lib.cn_conv3x3p_probe.argtypes = [ctypes.c_int, ctypes.c_void_p]
NAMES = {0: "full", 1: "no MFMA", 2: "no frag reads / MFMA", 12: "no DMA", 16: "no epilogue", 32: "no stores",
         28: "no DMA, no epilogue", 3 | 16: "barriers + DMA only", 31: "barriers only", 64: "half the frag reads",
         64 | 16: "half the frag reads, no epilogue"}
print("== probes (instrumented instantiation), B = 32")
for (ci, H, W, co) in ([(64, 128, 128, 64), (256, 32, 32, 256)] if os.environ.get("PROBE") else []):
    for res in (0, 1):
        c = Case(32, ci, H, W, co, res=res)
        fl = 2.0 * 32 * H * W * co * ci * 9
        lib.cn_set_tuning(28, 1)
        prof = torch.zeros((512, 6, 8), device=dev, dtype=torch.int64)
        row = []
        for dbg in (0, 64, 0, 64, 64 | 16, 1, 2, 12, 16, 32, 28, 19, 31):
            lib.cn_conv3x3p_probe(dbg, prof.data_ptr())
            ms = c.time(10)
            row.append("%s %.4f" % (NAMES[dbg], ms))
        print("%-20s res %d | %s" % (str((ci, H, W, co)), res, " | ".join(row)))
        lib.cn_conv3x3p_probe(0, prof.data_ptr())
        prof.zero_()
        c.launch()
        torch.cuda.synchronize()
        pr = prof.cpu().double()
        used = pr[:, 0, 3] > 0
        cons, lw, lh = pr[used][:, :4], pr[used][:, 4], pr[used][:, 5]
        print("   consumers: total %.0f cyc, epilogue %.0f, barrier wait %.0f, stages %.0f | weight loader: total %.0f, vmcnt wait %.0f, barrier wait %.0f | halo loader: vmcnt wait %.0f, barrier wait %.0f" % (
            cons[..., 0].mean(), cons[..., 1].mean(), cons[..., 2].mean(), cons[..., 3].mean(),
            lw[:, 0].mean(), lw[:, 1].mean(), lw[:, 2].mean(), lh[:, 1].mean(), lh[:, 2].mean()))
        # wall clock (100 MHz) of every workgroup: shader clock and how many workgroups run at a time
        rt0, rt1 = pr[used][:, 0, 4], pr[used][:, 0, 5]
        t0 = rt0.min()
        span = float(rt1.max() - t0) / 100.0        # us
        clk = (cons[:, 0, 0] / ((rt1 - rt0) * 10.0)).mean()   # cycles per ns
        late = int((rt0 > rt1.min()).sum())
        hw = pr[used][:, 0, 7].long()
        cu_id = (pr[used][:, 0, 6].long() << 16) | (((hw >> 8) & 0xf) << 4) | ((hw >> 13) & 0x7) << 8 | ((hw >> 16) & 0x3) << 12
        ncu = len(set(cu_id.tolist()))
        print("   launch span %.1f us; shader clock %.2f GHz; workgroups started after the first one ended: %d of %d; start spread %.1f us; distinct (xcc, se, sh, cu) ids %d; occupancy API: %d / %d" % (
            span, clk, late, int(used.sum()), float(rt0.max() - t0) / 100.0, ncu, lib.cn_conv3x3p_occupancy(0, 0), lib.cn_conv3x3p_occupancy(1, 0)))
        lib.cn_conv3x3p_probe(0, None)
        del c
