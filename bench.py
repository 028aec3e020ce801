#!/usr/bin/env python
"""Headline benchmark: images/s of the CenterNet hot path (network + decode) on 512x512
synthetic input.  Default = BASELINE configs[1]: ctdet ResNet-18-DCN, batch 32 per GPU, fp32.

    python bench.py                                   # 1 GPU, configs[1]
    python bench.py --gpus 8 --steps 50 --warmup 5    # spawns 8 ranks itself (torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # or launched by the driver

    --config 2|3|4 selects the other BASELINE configurations (dla_34 ctdet, dla_34 multi_pose,
    hourglass fp16 batch 8); --arch / --task / --batch / --fp16 override piecewise.

One step = detector.run_batch(images): backbone + DCN + heads + fused sigmoid / peak-NMS /
top-K decode over one device-resident batch -- the same entry point in warm-up and in the
timed loop.  Images shard over ranks (weak scaling, no collective in the loop); rank 0
broadcasts the flat weight buffer once at start-up over RCCL.  Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md)
# f32s: every fp32 product is three fp16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate), so the
# matrix-pipe ceiling in ALGORITHMIC (fp32-equivalent) FLOPs is a third of the fp16 peak
F32S_MFMA_PEAK_TF = F16_MFMA_PEAK_TF / 3.0

# kernels per decode call (cn_decode.hip): ctdet maps of <= 128 x 128 cells (a 512 x 512 input) take the
# one-launch form plane_select_merge_kernel, larger ones group_max + collect_merge
DECODE_LAUNCHES = {"ctdet": 1, "ctdet_large": 2, "multi_pose": 5}

# BASELINE.json configs[1..4] (configs[0] is the reference's CPU plumbing case: cpu_baseline)
CONFIGS = {
    1: dict(task="ctdet", arch="resdcn_18", batch=32, fp16=False),
    2: dict(task="ctdet", arch="dla_34", batch=32, fp16=False),       # 256 over 8 GPUs = 32/GPU
    3: dict(task="multi_pose", arch="dla_34", batch=32, fp16=False),
    4: dict(task="ctdet", arch="hourglass", batch=8, fp16=True),
}


def device_identity(index, stub=False):
    """What a rank says about the device it runs on: used to prove that N ranks sit on N GPUs."""
    if stub:
        # BENCH_FAKE_UUID: the shared-device refusal under test on a host without GPUs (tests/test_sharding.py)
        return {"rank": int(os.environ.get("RANK", "0")), "device": "cpu",
                "uuid": os.environ.get("BENCH_FAKE_UUID") or "cpu-%d" % os.getpid(), "name": "stub"}
    import torch
    pr = torch.cuda.get_device_properties(index)
    uuid = getattr(pr, "uuid", None)
    uuid = str(uuid) if uuid is not None else "%s/%s" % (getattr(pr, "pci_bus_id", "?"), getattr(pr, "pci_device_id", index))
    return {"rank": int(os.environ.get("RANK", "0")), "device": "cuda:%d" % index, "uuid": uuid,
            "name": pr.name, "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}


def kernel_tree_sha():
    """Fingerprint of everything a launch list depends on: the HIP sources, the C ABI header and the planner.
    (.git does not travel to the GPU box, so a commit hash cannot be read there; this digest is what
    tools/pmc_traffic.py records as a profile's `head` and what a run compares itself with.)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "centernet_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "centernet_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "centernet_amd.h"), os.path.join(ROOT, "centernet_amd", "engine.py")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(tag, check_head=True):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary of this
    configuration (tools/pmc_traffic.py: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), summed
    per kernel class; {} when no profile of this configuration is committed.  A profile whose recorded
    `head` (kernel_tree_sha at the time it was taken) differs from the running tree is REFUSED:
    {"_stale": (its head, this tree's)} -- the caller reports traffic null."""
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % tag)))
    if not hits and tag == "cfg1":
        hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not hits:
        return {}, None
    with open(hits[-1]) as f:
        raw = json.load(f)
    # a profile taken on OTHER kernels says nothing about this run's traffic: refuse it (traffic: null)
    head = (raw.get("_meta") or {}).get("head")
    if check_head and head != kernel_tree_sha():
        return {"_stale": (head, kernel_tree_sha())}, os.path.basename(hits[-1])
    cls = {"conv": 0.0, "dcn": 0.0, "decode": 0.0, "maxpool": 0.0}
    n = {"conv": 0, "dcn": 0, "decode": 0, "maxpool": 0}
    def short(name):
        # kernel names with _Float16 template arguments stay mangled in the rocprofv3 tables:
        # _ZN12_GLOBAL__N_112igemm_kernelIDF16_Li128E...Li0E... -> ("igemm_kernel", [128, ..., 0, ...])
        if name.startswith("_ZN"):
            import re
            m = re.match(r"_ZN\d+_GLOBAL__N_\d+([a-z_0-9]+?)(?:I|E)(.*)", name)
            if not m:
                return name, []
            return m.group(1), [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(2))]
        base = name.split("<")[0]
        args = [a.strip() for a in name[len(base):].strip("<>").split(",")] if "<" in name else []
        return base, [int(a) if a.lstrip("-").isdigit() else a for a in args[1:]]
    # profiled forward steps = launches of the fused-heads kernel in the arithmetic the run is
    # quoted in (exactly one per forward in every network; the fp32 calibration pass that precedes
    # an f32s run uses the <float> instantiation and is not counted -- its launches stay in the
    # class sums, a 2-in-36 overestimate with the profile command's 30 + 3 steps)
    def is_heads(k):
        base, targs = short(k)
        if base == "conv3x3p_kernel":     # <RES, OUT_PLAIN, DBG, HEADS, NTAP, S2>: the fused heads of the persistent kernel
            return len(targs) > 2 and targs[2] in ("true", 1)
        return base == "conv3x3s1_kernel" and len(targs) > 4 and targs[4] in ("true", 1)
    heads = {k: v.get("launches", 0) for k, v in raw.items() if isinstance(v, dict) and is_heads(k)}
    main = [n_ for k, n_ in heads.items() if "cn_f32s" in k] or [n_ for k, n_ in heads.items() if "DF16_" in k] \
        or list(heads.values())
    steps = sum(main)
    if not steps:   # networks without fused heads: the stem kernels (one per forward)
        steps = sum(v.get("launches", 0) for k, v in raw.items()
                    if isinstance(v, dict) and short(k)[0].startswith("stem_"))
    steps = max(1, steps or raw.get("nms_topk_kernel", {}).get("launches", 1))
    for k, v in raw.items():
        if not isinstance(v, dict) or "hbm_bytes_per_launch" not in v:
            continue
        base, targs = short(k)
        if base == "igemm_kernel":
            c = "dcn" if len(targs) > 4 and targs[4] in (2, 3) else "conv"   # AMODE
        elif base.startswith(("dcn_reg_kernel", "dcn_win_kernel", "dcn_team_kernel", "dcn_wide_kernel")):   # LDS-window forms (cn_dcn2 / 3 / 4.hip)
            c = "dcn"
        elif base.startswith(("stem_", "splitk_reduce", "conv3x3", "conv16_kernel", "heads_", "offconv_kernel", "proj1x1_kernel")):
            c = "conv"
        elif base.startswith(("nms_topk", "merge_topk", "peak_", "group_", "pose_match", "decode_", "collect_merge",
                              "plane_select_merge")):
            c = "decode"
        elif base.startswith("maxpool"):
            c = "maxpool"
        else:
            continue
        cls[c] += v["hbm_bytes_per_launch"] * v["launches"] / steps   # bytes per forward step
        n[c] += int(round(v["launches"] / steps))
    return {c: (cls[c], n[c]) for c in cls if n[c]}, os.path.basename(hits[-1])


def rocprof_kernel_for(avg_us, cfg):
    """The kernel of the newest committed `rocprofv3 --kernel-trace --stats` summary of this
    configuration whose average duration is closest to ``avg_us`` (and within 20 %): name + its
    rocprof figures, so the line names a kernel, not only a class."""
    import csv
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats_*cfg%d.csv" % cfg)))
    if not hits:
        return None
    best = None
    with open(hits[-1]) as f:
        for row in csv.DictReader(f):
            try:
                us = float(row["AverageNs"]) / 1e3
            except (KeyError, ValueError):
                continue
            if best is None or abs(us - avg_us) < abs(best[0] - avg_us):
                best = (us, row)
    if best is None or abs(best[0] - avg_us) > 0.2 * avg_us:
        return None
    name = best[1]["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return {"name": name, "rocprof_avg_us": best[0], "rocprof_calls": int(best[1]["Calls"]),
            "rocprof_pct": float(best[1]["Percentage"]), "file": "profiles/" + os.path.basename(hits[-1])}


def dominant_launch(det, images, plan, steps, peak_tf, cfg):
    """The single longest launch of the step: HIP events after EVERY launch over a few extra
    steps (outside the timed region: a marker per launch costs ~7 us of bubble each)."""
    import torch
    metas, trace = plan.b.meta, plan.b.trace
    acc = [0.0] * len(metas)
    for _ in range(steps):
        probe = {"event_after": set(range(len(metas)))}
        det.run_batch(images, probe=probe)
        torch.cuda.synchronize()
        ev = probe["net_events"]
        for i in range(len(metas)):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    i = max(range(len(metas)), key=lambda j: acc[j])
    ms = acc[i] / steps
    kind, act = trace[i]
    tf = metas[i]["flops"] / ms / 1e9 if ms > 0 else 0.0
    out = {"op_index": i, "kind": kind, "out_shape_BHWC": [act.B, act.H, act.W, act.C] if act is not None else None,
           "avg_ms": ms, "share_of_step": ms / (sum(acc) / steps), "achieved_TFLOPs": tf,
           "frac_of_peak": tf / peak_tf if peak_tf else None,
           "algorithmic_bytes": metas[i]["bytes"], "rocprof": rocprof_kernel_for(ms * 1e3, cfg)}
    return out


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                   help="BASELINE.json configs[i]")
    p.add_argument("--task", default=None, choices=["ctdet", "multi_pose"])
    p.add_argument("--arch", default=None)
    p.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    p.add_argument("--res", type=int, default=512)
    p.add_argument("--fp16", action="store_true", default=None,
                   help="fp16 activations/weights, fp32 accumulate (configs[4], hourglass only)")
    p.add_argument("--fp32-mfma", action="store_true",
                   help="compute on v_mfma_f32_32x32x2_f32 (the round-1 kernels) instead of f32s")
    p.add_argument("--tune", action="append", default=[],
                   help="KEY=VALUE for cn_set_tuning (A/B experiments; not used by the driver)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-fp32-leg", action="store_true",
                   help="skip the short fp32-MFMA leg that is reported next to the f32s headline")
    p.add_argument("--cpu-seconds", type=float, default=5.0,
                   help="bound of the CPU-baseline sample (seconds of CPU work)")
    p.add_argument("--per-op", action="store_true", help="print per-launch timings to stderr")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the short runs of BASELINE configs[2..4] behind the headline loop")
    p.add_argument("--secondary-steps", type=int, default=20)
    a = p.parse_args(argv)
    cfg = CONFIGS[a.config]
    for k, v in cfg.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def precision_check(dev):
    """One dense layer (128 -> 128, 3x3, 2 x 64 x 64) on both compute modes against torch fp64,
    at three activation scales: max |error| relative to the output rms.  f32s (three fp16 MFMAs
    per fp32 product, fp32 accumulate, tensors stored with a per-tensor power-of-two exponent) is
    held to the accuracy of the plain fp32 matrix instruction at every scale."""
    import torch
    import torch.nn.functional as F
    from centernet_amd.engine import PlanBuilder, Act, exponent_for
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn((2, 128, 64, 64), generator=g).relu_()
    w = torch.randn((128, 128, 3, 3), generator=g) * (2.0 / (128 * 9)) ** 0.5
    out = {}
    for sname, sc in (("", 1.0), ("_x1e-4", 1e-4), ("_x1e+4", 1e4)):
        x = x0 * sc
        ref = F.conv2d(x.double(), w.double(), padding=1).permute(0, 2, 3, 1)
        rms = float(ref.pow(2).mean().sqrt())
        exps = {"x": exponent_for(float(x.abs().max())), "t1": exponent_for(float(ref.abs().max()))}
        for name, split in (("f32s", True), ("fp32_mfma", False)):
            pb = PlanBuilder(dev, 2, 64, 64, split=split, exps=exps)
            xa = Act(x.permute(0, 2, 3, 1).contiguous().to(dev), 2, 64, 64, 128, exp=pb._exp("x"), lid="x")
            y = pb.plain(pb.conv(xa, w, stride=1, padding=1))
            for op in pb.ops:
                op()
            torch.cuda.synchronize()
            out[name + "_max_err_over_rms" + sname] = float((y.t.double().cpu() - ref).abs().max()) / rms
    out["reference"] = "torch fp64 conv2d"
    return out


def box_calibration(dev, target_ms=50.0):
    """What THIS box delivers right now, measured just before the timed region: a register-only
    v_mfma_f32_32x32x16_f16 loop on every SIMD (no memory traffic: the matrix pipe at the clock
    the part holds under load) and a float4 copy of 512 MiB (twice the Infinity Cache; read +
    write bytes over time), each sized to ~``target_ms`` from a short trial launch.  The headline
    divided by these two says whether a change between runs is the code or the box."""
    import torch
    from centernet_amd import native
    lib, st = native.lib(), native.stream_ptr
    sink = torch.zeros(16, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def mfma(iters):
        e0.record()
        fl = lib.cn_calib_mfma_f16(native.ptr(sink), iters, st())
        e1.record()
        torch.cuda.synchronize()
        assert fl > 0, "cn_calib_mfma_f16 failed"
        return fl, e0.elapsed_time(e1)
    mfma(200)                                   # lazy code load
    fl, ms = mfma(2000)
    iters = max(2000, int(2000 * target_ms / max(ms, 1e-3)))
    fl, ms = mfma(iters)
    out = {"mfma_f16_TFLOPs": fl / ms / 1e9, "mfma_ms": ms,
           "mfma_frac_of_2500": fl / ms / 1e9 / F16_MFMA_PEAK_TF}
    nbytes = 512 << 20
    src = torch.empty(nbytes // 4, device=dev).normal_()
    dst = torch.empty_like(src)

    def copy(n):
        e0.record()
        for _ in range(n):
            native.check(lib.cn_calib_copy(native.ptr(src), native.ptr(dst), nbytes, st()), "cn_calib_copy")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    copy(2)
    ms = copy(4)
    n = max(4, int(4 * target_ms / max(ms, 1e-3)))
    ms = copy(n)
    out.update({"copy_TBs": 2.0 * nbytes * n / ms / 1e9, "copy_ms": ms,
                "copy_frac_of_8TBs": 2.0 * nbytes * n / ms / 1e6 / HBM_PEAK_GBS,
                "what": "register-only v_mfma_f32_32x32x16_f16 loop (1024 workgroups x 4 waves) and a float4 "
                        "copy of 512 MiB (read + write bytes), ~%d ms each, right before the timed region" % target_ms})
    del src, dst
    # latency: dependent loads through 512 MiB (beyond L2 and the Infinity Cache: HBM) and through
    # 256 KiB (L2), dependent device-scope atomics -- ONE lane, 4096 steps each
    steps = 4096

    def walk(n_elems, stride, fresh=False):
        chain = torch.arange(n_elems, device=dev, dtype=torch.int32) + stride
        chain = torch.where(chain >= n_elems, chain - n_elems, chain).contiguous()
        atom = torch.zeros(64, device=dev, dtype=torch.int32)
        res = torch.zeros(4, device=dev, dtype=torch.int64)
        best = None
        for _ in range(3):
            # (fresh: every repeat walks lines no earlier walk touched -- a repeat of the same walk
            # would find its 4096 lines in L2)
            native.check(lib.cn_calib_latency(native.ptr(chain), 32 * _ if fresh else 0, steps, native.ptr(atom),
                                              native.ptr(res), st()), "cn_calib_latency")
            torch.cuda.synchronize()
            r = res.cpu().tolist()
            cur = (r[0] * 10.0 / steps, r[1] * 10.0 / steps)      # 100 MHz ticks -> ns per step
            best = cur if best is None else (min(best[0], cur[0]), min(best[1], cur[1]))
        return best
    hbm_ns, atomic_ns = walk((1 << 30) // 4, 4099 * 64 + 32, fresh=True)   # a new 128-byte line 1 MiB away, every step, over 1 GiB
    l2_ns, _ = walk((1 << 20) // 4, 37 * 32)                     # 1 MiB: beyond L1, inside the XCD's 4 MiB L2
    l1_ns, _ = walk((8 << 10) // 4, 5 * 32)                      # 8 KiB: L1
    out.update({"latency_ns": {"hbm_dependent_load": hbm_ns, "l2_dependent_load": l2_ns, "l1_dependent_load": l1_ns,
                               "device_scope_atomic_round_trip": atomic_ns,
                               "what": "one lane, %d dependent steps each (s_memrealtime, 100 MHz): walks over "
                                       "1 GiB, 1 MiB and 8 KiB, atomic additions on one word" % steps}})
    return out


def _time_images(fn, make_input, seconds):
    fn(make_input(0))   # warm-up
    t0 = time.time()
    done = 0
    while (time.time() - t0) < seconds:
        fn(make_input(1 + done))
        done += 1
    return done, time.time() - t0


def cpu_baseline(task, arch, state_dict, heads, res, seconds):
    """CPU leg, timed on this host's cores in the same run.

    kind "port": oracle/net_oracle (torch-CPU dense ops -- the reference's own third-party
    arithmetic -- + C DCNv2 + C decode) on the SAME architecture; the reference's DCNv2 has no
    CPU implementation (DCNv2/src/dcn_v2.c:5-16 only prints), so a port is the only CPU form of
    the DCN networks.  Where /root/reference is present (build container; never on the GPU box)
    the reference's OWN msra_resnet.PoseResNet + models/decode.ctdet_decode (BASELINE
    configs[0]: ResNet-18 without DCN, batch 1) is timed beside it as `reference_res18`."""
    import torch
    from centernet_amd import synth
    from oracle import net_oracle, cref
    cref.lib()
    cores = min(os.cpu_count() or 1, 64)   # more threads than this slows batch-1 convs down
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    if task == "ctdet":
        fn = lambda x: net_oracle.ctdet_process(arch, state_dict, x, heads)      # noqa: E731
    else:
        fn = lambda x: net_oracle.multi_pose_process(arch, state_dict, x, heads)  # noqa: E731
    done, dt = _time_images(fn, lambda i: synth.images(1, res, res, seed=1 + i), seconds)
    out = {"value": done / dt, "unit": "img/s", "cores": cores, "kind": "port",
           "sample": "%d images %dx%d batch 1 in %.1f s: oracle/net_oracle %s %s -- a PORT (torch-CPU "
                     "convs + C DCNv2 + C decode), not the reference's code: the reference has no "
                     "CPU DCNv2" % (done, res, res, dt, task, arch)}
    # the reference's OWN CPU path (BASELINE configs[0]) as recorded in the build container --
    # /root/reference does not exist on the GPU box, so it cannot be timed in this run there
    rec = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_reference_res18.json")))
    if rec:
        with open(rec[-1]) as f:
            out["reference_res18_recorded"] = dict(
                json.load(f), file="profiles/" + os.path.basename(rec[-1]), replay=True,
                replay_note="NOT timed in this run: the record of an earlier run in the build container, where "
                            "/root/reference exists (it does not on the GPU box); `reference_res18`, when present, IS "
                            "this run's timing of the same code")
    ref_src = "/root/reference/src/lib"
    if os.path.isdir(ref_src):
        try:
            out["reference_res18"] = _reference_cpu_res18(ref_src, res, min(seconds, 10.0), cores)
        except Exception as e:   # the reference tree is optional, never fatal
            out["reference_res18"] = {"error": repr(e)[:200]}
    return out


def _reference_cpu_res18(ref_src, res, seconds, cores):
    """The reference's own CPU PyTorch path: msra_resnet.PoseResNet(BasicBlock,
    [2,2,2,2]) -> sigmoid_ -> models/decode.ctdet_decode at batch 1, imported from
    /root/reference (no bytecode written there)."""
    import importlib
    import torch
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref_src)
    try:
        mr = importlib.import_module("models.networks.msra_resnet")
        dec = importlib.import_module("models.decode")
    finally:
        sys.path.remove(ref_src)
    heads = {"hm": 80, "wh": 2, "reg": 2}
    torch.manual_seed(317)
    m = mr.PoseResNet(mr.BasicBlock, [2, 2, 2, 2], heads, head_conv=64).eval()

    def fn(x):
        with torch.no_grad():
            o = m(x)[-1]
            return dec.ctdet_decode(o["hm"].sigmoid_(), o["wh"], reg=o["reg"], K=100)
    g = torch.Generator().manual_seed(0)
    done, dt = _time_images(fn, lambda i: torch.randn((1, 3, res, res), generator=g), seconds)
    return {"value": done / dt, "unit": "img/s", "cores": cores, "kind": "reference",
            "sample": "%d images %dx%d batch 1 in %.1f s: reference msra_resnet res_18 (no DCN) + "
                      "models/decode.ctdet_decode, torch CPU" % (done, res, res, dt)}


def secondary_config(cfg_id, res, steps, dev):
    """BASELINE configs[2..4] in the driver-timed line: the SAME step as the headline (det.run_batch on a
    device-resident batch, range words read inside the timed region) for a few steps on a fresh detector;
    value, ms_per_step and the per-class roofline fractions from HIP events at the class boundaries."""
    import contextlib
    import torch
    from centernet_amd import synth
    from centernet_amd.opts import opts
    from centernet_amd.detectors import detector_factory
    c = CONFIGS[cfg_id]
    opt = opts().init([c["task"], "--arch", c["arch"], "--input_res", str(res)])
    with contextlib.redirect_stdout(sys.stderr):
        det = detector_factory[opt.task](opt)
    synth.fill_state_dict_(det.model, 317)
    det.model.invalidate_plans()
    if c["fp16"]:
        det.model.half_compute()
    B = c["batch"]
    images = synth.images(B, res, res, seed=100).to(dev)
    for _ in range(10):      # (a fresh detector's first steps carry plan building, workspace growth and lazy code loads)
        det.run_batch(images)
    torch.cuda.synchronize()
    if not c["fp16"]:
        assert det.range_ok(images), "f32s range check failed during warm-up (config %d)" % cfg_id
    plan = det.model.plan_for(B, res, res, dev)
    metas = plan.b.meta
    nops = len(metas)
    bounds = [i for i in range(nops) if i == nops - 1 or metas[i]["kind"] != metas[i + 1]["kind"]]
    seg_first = [0] + [b + 1 for b in bounds[:-1]]
    probes = []
    t0 = time.perf_counter()
    for _ in range(steps):
        probe = {"event_after": set(bounds)}
        det.run_batch(images, probe=probe)
        probes.append(probe)
    clean = det.range_ok() if not c["fp16"] else True
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kinds = {}
    for pr in probes:
        evs = pr["net_events"]
        for si, (first, last) in enumerate(zip(seg_first, bounds)):
            k = kinds.setdefault(metas[first]["kind"], {"ms": 0.0, "flops": 0, "launches": 0})
            k["ms"] += evs[si].elapsed_time(evs[si + 1])
            for m in metas[first:last + 1]:
                k["flops"] += m["flops"]; k["launches"] += 1
    dec_ms = sum(pr["dec_events"][0].elapsed_time(pr["dec_events"][1]) for pr in probes)
    Ho = res // 4
    if c["task"] == "ctdet":
        dec_bytes = B * (opt.num_classes * Ho * Ho * 4 + 2 * 2 * Ho * Ho * 4 + opt.K * 6 * 4)
    else:
        dec_bytes = B * ((1 + 2 + 34 + 2 + 17 + 2) * Ho * Ho * 4 + opt.K * 40 * 4)
    peak = F16_MFMA_PEAK_TF if c["fp16"] else F32S_MFMA_PEAK_TF
    classes = {}
    for k, v in kinds.items():
        sec = v["ms"] * 1e-3
        if v["flops"] > 0 and sec > 0:
            classes[k] = {"bound": "mfma", "achieved_TFLOPs": v["flops"] / sec / 1e12,
                          "frac": v["flops"] / sec / 1e12 / peak, "launches_per_step": v["launches"] // steps,
                          "time_share": round(v["ms"] / (dt * 1e3), 4)}
        else:
            classes[k] = {"bound": "hbm", "launches_per_step": v["launches"] // steps,
                          "time_share": round(v["ms"] / (dt * 1e3), 4)}
    classes["decode"] = {"bound": "hbm", "achieved_GBs": dec_bytes * steps / (dec_ms * 1e-3) / 1e9 if dec_ms else 0.0,
                         "frac": (dec_bytes * steps / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if dec_ms else 0.0,
                         "launches_per_step": DECODE_LAUNCHES[c["task"]], "time_share": round(dec_ms / (dt * 1e3), 4)}
    out = {"config": "%s %s %dx%d, batch %d (BASELINE configs[%d])" % (c["task"], c["arch"], res, res, B, cfg_id),
           "value": B * steps / dt, "unit": "img/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "dtype": "f16" if c["fp16"] else "f32s", "peak_TFLOPs": peak, "range_clean": bool(clean),
           "gflop_per_image": plan.flops / B / 1e9, "classes": classes}
    del det, plan, images
    torch.cuda.empty_cache()
    return out


def cross_rank_agreement(per_rank, tol=1e-4):
    """Detections of ONE shared batch from every rank against rank 0's.  (B, K, 6) rows are
    [x1, y1, x2, y2, score, class]: the sorted scores must agree within ``tol`` (rank by rank --
    robust against two near-tied rows swapping places) and rows must agree in place (same class,
    boxes within 1e-2 px) except for such swaps; any other tensor: max |difference| <= tol."""
    import torch
    ref = per_rank[0]
    out = {"ranks": len(per_rank), "tol": tol, "max_score_diff": 0.0, "rows_equal_in_place": 1.0, "ok": True}
    for t in per_rank[1:]:
        if ref.dim() == 3 and ref.shape[-1] >= 6:
            ds = float((t[..., 4] - ref[..., 4]).abs().max())
            same = ((t[..., 5] == ref[..., 5]) & ((t[..., :4] - ref[..., :4]).abs().amax(-1) <= 1e-2)).float().mean()
            out["max_score_diff"] = max(out["max_score_diff"], ds)
            out["rows_equal_in_place"] = min(out["rows_equal_in_place"], float(same))
        else:
            out["max_score_diff"] = max(out["max_score_diff"], float((t - ref).abs().max()))
    out["ok"] = out["max_score_diff"] <= tol and out["rows_equal_in_place"] >= 0.98
    return out


class _StubEvent(object):
    """HIP-event stand-in of the stub run (BENCH_STUB=1): wall-clock stamps."""

    def __init__(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _StubDetector(object):
    """BENCH_STUB=1 (tests/test_sharding.py): the detector replaced by a few CPU matmuls so that
    the whole rank logic of this file -- process group, weight broadcast, barriers, max-over-ranks
    timing, rank 0's JSON line -- runs under gloo on a host without a GPU.  The line says
    "data": "stub"; nothing in it is a measurement of the product."""

    class _Plan(object):
        def __init__(self):
            from types import SimpleNamespace as NS
            act = NS(B=1, H=1, W=1, C=1)
            self.b = NS(meta=[dict(kind="conv", flops=2 * 64 ** 3, bytes=3 * 4 * 64 * 64)],
                        trace=[("conv", act)])
            self.flops = 2 * 64 ** 3

    def __init__(self, opt, dev):
        import torch
        self.opt = opt
        self.model = torch.nn.Linear(64, 64).to(dev)
        self.model.invalidate_plans = lambda: None
        self.model.plan_for = lambda *a: self._Plan()
        self.model.half_compute = self.model.fp32_mfma = lambda *a: None

    def run_batch(self, images, probe=None):
        import torch
        e = [_StubEvent()]
        with torch.no_grad():
            y = self.model(images.reshape(-1)[:64 * 64].reshape(64, 64))
        e.append(_StubEvent())
        if probe is not None:
            probe["net_events"] = e
            probe["dec_events"] = (e[1], _StubEvent())
        return y

    def range_ok(self, images=None):
        return True


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves (one process per GPU)
        from centernet_amd.sharding import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    # BENCH_STUB=1: CPU + gloo + _StubDetector -- the rank logic below under test on a host
    # without a GPU (tests/test_sharding.py); the driver never sets it
    stub = os.environ.get("BENCH_STUB") == "1"
    assert stub or torch.cuda.is_available(), "bench.py needs a HIP device"
    # BENCH_DEVICE / BENCH_BACKEND exist only so the multi-rank control flow can be exercised on
    # a single-GPU box (two ranks sharing cuda:0 over gloo); the driver never sets them.
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    backend = "gloo" if stub else os.environ.get("BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None   # noqa: E731
    else:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        sync = torch.cuda.synchronize
    dist = None
    world = 1
    if env_world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=env_world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=env_world)
        except Exception as e:
            raise SystemExit(
                "bench.py: rank %d/%d could not join the %s process group on %s (%s: %s). Check: one "
                "visible GPU per rank (LOCAL_RANK=%d -> cuda:%d of %d), MASTER_ADDR=127.0.0.1 and a free "
                "MASTER_PORT, HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL needs dmabuf IPC on this driver)."
                % (rank, env_world, backend, dev, type(e).__name__, str(e)[:200], local_rank, dev_index,
                   torch.cuda.device_count() if not stub else 0))
        world = dist.get_world_size()
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s); launch with "
                         "`python bench.py --gpus N` (self-spawning) or torchrun --nproc-per-node N"
                         % (a.gpus, world))
    # first contact with a multi-GPU node must not be able to pass silently on ONE device: every rank
    # reports the device it really sits on, rank 0 refuses a world in which two ranks share one
    rank_devices = None
    if dist is not None:
        mine = device_identity(dev_index, stub)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
        ids = [d["uuid"] for d in rank_devices]
        if len(set(ids)) != len(ids) and os.environ.get("BENCH_ALLOW_SHARED_DEVICE") != "1":
            dist.destroy_process_group()
            raise SystemExit("bench.py: %d ranks but only %d distinct devices %r -- two ranks share a GPU "
                             "(check LOCAL_RANK / HIP_VISIBLE_DEVICES); refusing to report a scaling number"
                             % (world, len(set(ids)), ids))

    from centernet_amd import synth, native
    for kv in a.tune:
        k, v = kv.split("=")
        assert native.lib().cn_set_tuning(int(k), int(v)) == 0, kv
    from centernet_amd.opts import opts
    from centernet_amd.sharding import broadcast_weights

    opt = opts().init([a.task, "--arch", a.arch, "--input_res", str(a.res)])
    import contextlib
    if stub:
        torch.manual_seed(rank)            # ranks start with DIFFERENT weights
        det = _StubDetector(opt, dev)
    else:
        from centernet_amd.detectors import detector_factory
        with contextlib.redirect_stdout(sys.stderr):   # the detector prints 'Creating model...'
            det = detector_factory[opt.task](opt)
    if rank == 0 and not stub:
        synth.fill_state_dict_(det.model, 317)
    bcast_bytes, bcast_ms = 0, None
    if world > 1:
        sync()
        tb = time.perf_counter()
        bcast_bytes = broadcast_weights(det.model, src=0)   # ONE flat RCCL broadcast, then none
        sync()
        bcast_ms = (time.perf_counter() - tb) * 1e3
    det.model.invalidate_plans()
    if a.fp16:
        det.model.half_compute()
    if a.fp32_mfma:
        det.model.fp32_mfma()
    B = a.batch
    images = synth.images(B, a.res, a.res, seed=100 + rank).to(dev)

    for _ in range(max(a.warmup, 1)):
        dets = det.run_batch(images)
    sync()
    # f32s: the first forward calibrated the per-tensor exponents on this batch; a clamped value
    # here would mean the calibration is broken -- never time a run that is not range-clean
    assert det.range_ok(images), "f32s range check failed during warm-up"
    agreement = None
    if dist is not None:
        # every rank calibrated its f32s exponents on ITS OWN images: one shared batch through all
        # replicas must still give the same detections (scores within 1e-4) behind the broadcast
        shared = synth.images(B, a.res, a.res, seed=99).to(dev)
        mine = det.run_batch(shared).detach().float().contiguous().clone()
        sync()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        agreement = cross_rank_agreement([g.cpu() for g in gathered])
        assert agreement["ok"], "replicas disagree on a shared batch: %r" % (agreement,)
        assert det.range_ok(shared), "f32s range check failed on the shared batch"
    calib = box_calibration(dev) if (rank == 0 and not stub) else None
    if dist is not None:
        dist.barrier()
    sync()

    plan = det.model.plan_for(B, a.res, a.res, dev)
    # HIP events on the launch stream, inside the timed region.  A marker after every launch
    # costs ~7 us of bubble each, so by default events sit only where the kernel class changes
    # (the per-class sums stay exact); --per-op records one after every launch.
    metas = plan.b.meta
    nops = len(metas)
    if a.per_op:
        bounds = list(range(nops))
    else:
        bounds = [i for i in range(nops) if i == nops - 1 or metas[i]["kind"] != metas[i + 1]["kind"]]
    bset = set(bounds)
    seg_first = [0] + [b + 1 for b in bounds[:-1]]   # first op of every segment
    probes = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        probe = {"event_after": bset}
        dets = det.run_batch(images, probe=probe)    # same entry point as the warm-up
        probes.append(probe)
    # inside the timed region: the (synchronising) look at the f32s range words of all K steps
    range_clean = det.range_ok()
    sync()
    dt_own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    per_rank_rate = None
    if dist is not None:
        # every rank's own rate (its loop without the closing barrier), so that a straggler GPU
        # shows in the one line; the headline uses the MAX time over ranks
        mine = torch.tensor([B * a.steps / dt_own], device=dev, dtype=torch.float64)
        rates = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rates, mine)
        per_rank_rate = [round(float(r.item()), 1) for r in rates]
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if not range_clean:
        raise SystemExit("bench.py: an f32s forward clamped a value inside the timed region")
    clock = None
    if rank == 0 and not stub:
        # the shader clock WHILE the step runs: one spinning lane on a side stream counts core cycles
        # (s_memtime) against the constant 100 MHz clock for about two steps, with untimed steps of the
        # same loop on the launch stream around it -- and the same probe on the resting part.  What power
        # management does to THIS box under THIS kernel mix; the register-only matrix loop of
        # box_calibration does not pull the clock down the way the real mix does.
        from centernet_amd import native
        lib = native.lib()
        cres = torch.zeros(4, device=dev, dtype=torch.int64)
        side = torch.cuda.Stream(device=dev)
        step_ticks = max(3000, int(2.0 * dt / a.steps * 1e8))      # two steps, in 10 ns ticks

        def probe(ticks, stream):
            native.check(lib.cn_calib_clock(native.ptr(cres), ticks, ctypes.c_void_p(stream.cuda_stream)),
                         "cn_calib_clock")
        for _ in range(3):
            det.run_batch(images)
        probe(step_ticks, side)                 # (no event between the streams: the probe starts now)
        for _ in range(5):
            det.run_batch(images)
        torch.cuda.synchronize()
        c, t = cres.cpu().tolist()[:2]
        loaded = 100.0 * c / max(t, 1)
        time.sleep(0.5)
        probe(3000, torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        c, t = cres.cpu().tolist()[:2]
        rested = 100.0 * c / max(t, 1)
        # the same for the memory system: the dependent-load walk over 1 GiB and the device-scope atomic
        # chain of box_calibration, on the side stream WHILE steps run (fresh lines; ~2 ms of one lane)
        n_el = (1 << 30) // 4
        chain = torch.arange(n_el, device=dev, dtype=torch.int32) + (4099 * 64 + 32)
        chain = torch.where(chain >= n_el, chain - n_el, chain).contiguous()
        atom = torch.zeros(64, device=dev, dtype=torch.int32)
        lres = torch.zeros(4, device=dev, dtype=torch.int64)
        torch.cuda.synchronize()
        for _ in range(3):
            det.run_batch(images)
        native.check(lib.cn_calib_latency(native.ptr(chain), 32 * 7, 4096, native.ptr(atom), native.ptr(lres),
                                          ctypes.c_void_p(side.cuda_stream)), "cn_calib_latency")
        for _ in range(5):
            det.run_batch(images)
        torch.cuda.synchronize()
        lr = lres.cpu().tolist()
        del chain
        clock = {"core_MHz_while_the_step_runs": loaded, "core_MHz_at_rest": rested,
                 "hbm_dependent_load_ns_while_the_step_runs": lr[0] * 10.0 / 4096,
                 "device_scope_atomic_ns_while_the_step_runs": lr[1] * 10.0 / 4096,
                 "what": "s_memtime cycles of one spinning lane over two steps (side stream, steps running "
                         "on the launch stream) / over 30 us after 0.5 s of rest, against the constant 100 MHz clock"}
    fp32_leg = None
    if rank == 0 and not a.fp16 and not a.fp32_mfma and not a.no_fp32_leg and not stub:
        # the same step on the plain fp32 matrix instruction (v_mfma_f32_32x32x2_f32), a few
        # steps, so that the headline is never ambiguous about what the f32s arithmetic buys
        det.model.fp32_mfma(True)
        for _ in range(2):
            det.run_batch(images)
        torch.cuda.synchronize()
        n_leg = max(3, min(10, a.steps))
        t1 = time.perf_counter()
        for _ in range(n_leg):
            det.run_batch(images)
        torch.cuda.synchronize()
        leg_dt = time.perf_counter() - t1
        fp32_leg = {"value": B * n_leg / leg_dt, "unit": "img/s (this rank)", "steps": n_leg,
                    "ms_per_step": leg_dt / n_leg * 1e3, "dtype": "f32"}
        det.model.fp32_mfma(None)
    track_leg = None
    if rank == 0 and not a.fp16 and not a.fp32_mfma and not a.no_fp32_leg and not stub:
        # the same step with the range words compiled out of the plan (NULL range pointers: no
        # running maxima, no per-wave atomics, no fold launch) against the tracked plan, back to
        # back in this process: what "nothing saturates silently" costs on this box
        def leg(n):
            for _ in range(3):
                det.run_batch(images)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                det.run_batch(images)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3
        n_leg = max(5, min(30, a.steps))
        det.model.range_tracking(False)
        off_ms = leg(n_leg)
        det.model.range_tracking(True)
        on_ms = leg(n_leg)
        track_leg = {"value": B / off_ms * 1e3, "unit": "img/s (this rank)", "steps": n_leg, "ms_per_step": off_ms,
                     "tracked_ms_per_step_same_loop": on_ms, "tracking_cost_frac": (on_ms - off_ms) / on_ms,
                     "note": "untracked plans clamp silently: measurement only"}
    if rank == 0:
        plan = det.model.plan_for(B, a.res, a.res, dev)     # (the legs above rebuilt the plans)
        # ---- per-kernel-class time from the HIP events recorded inside the timed region
        kinds = {}
        for pr in probes:
            evs = pr["net_events"]
            for si, (first, last) in enumerate(zip(seg_first, bounds)):
                ms = evs[si].elapsed_time(evs[si + 1])
                k = kinds.setdefault(metas[first]["kind"], {"ms": 0.0, "flops": 0, "bytes": 0, "launches": 0})
                k["ms"] += ms
                for m in metas[first:last + 1]:
                    k["flops"] += m["flops"]; k["bytes"] += m["bytes"]; k["launches"] += 1
        if a.per_op:
            for i, m in enumerate(plan.b.meta):
                ms = sum(pr["net_events"][i].elapsed_time(pr["net_events"][i + 1]) for pr in probes) / a.steps
                kind, act = plan.b.trace[i]
                if act is None:
                    continue
                print("op %2d %-7s out(B,H,W,C)=(%d,%d,%d,%d) %8.3f ms  %7.2f TF/s  %7.1f GB/s" % (
                    i, kind, act.B, act.H, act.W, act.C, ms, m["flops"] / ms / 1e9 if ms else 0,
                    m["bytes"] / ms / 1e6 if ms else 0), file=sys.stderr)
        dec_ms = sum(pr["dec_events"][0].elapsed_time(pr["dec_events"][1]) for pr in probes)
        Ho = a.res // 4
        if a.task == "ctdet":   # hm + wh + reg read once, (K,6) written
            dec_bytes = B * (opt.num_classes * Ho * Ho * 4 + 2 * 2 * Ho * Ho * 4 + opt.K * 6 * 4)
        else:                   # hm 1 + wh 2 + hps 34 + reg 2 + hm_hp 17 + hp_offset 2, (K,40)
            dec_bytes = B * ((1 + 2 + 34 + 2 + 17 + 2) * Ho * Ho * 4 + opt.K * 40 * 4)
        kinds["decode"] = {"ms": dec_ms, "flops": 0, "bytes": dec_bytes * a.steps,
                           "launches": DECODE_LAUNCHES["ctdet_large" if a.task == "ctdet" and Ho > 128
                                                       else a.task] * a.steps}
        dom = max(kinds, key=lambda k: kinds[k]["ms"])
        standard = (a.res == 512 and not a.tune and not a.fp32_mfma and
                    all(getattr(a, k) == v for k, v in CONFIGS[a.config].items()))
        pmc, pmc_file = pmc_traffic("cfg%d" % a.config) if standard else ({}, None)
        pmc_stale = pmc.pop("_stale", None)

        def traffic(k):
            # measured HBM bytes per launch of this kernel class (committed PMC profile of the
            # same command), or null when no profile covers this configuration
            if k not in pmc:
                return None
            byts, launches = pmc[k]
            return byts / max(launches, 1)

        def roof(k, bound):
            s = kinds[k]
            sec = s["ms"] * 1e-3
            if bound == "mfma":
                ach = s["flops"] / sec / 1e12 if sec > 0 else 0.0
                peak = F16_MFMA_PEAK_TF if a.fp16 else (F32_MFMA_PEAK_TF if a.fp32_mfma else F32S_MFMA_PEAK_TF)
                unit = "TFLOP/s"
            else:
                ach = s["bytes"] / sec / 1e9 if sec > 0 else 0.0
                peak, unit = HBM_PEAK_GBS, "GB/s"
            tr, alg = traffic(k), s["bytes"] / max(s["launches"], 1)
            note = None
            if tr is not None and tr < alg:
                note = ("measured HBM bytes below the algorithmic count: " + (
                    "the algorithmic figure charges every decode input map as read once (SURVEY 8d), "
                    "but only the heat-maps are streamed -- wh / reg / hps / hp_offset are gathered at the "
                    "K selected cells (a few 32-byte sectors each)" if k == "decode" else
                    "consecutive launches hand small tensors and the weights over through L2 / the 256 MB "
                    "Infinity Cache (written, not yet evicted, read again), so fewer bytes reach HBM than a "
                    "per-launch count of inputs + outputs assumes"))
            return {"kernel": k, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                    "frac": ach / peak, "traffic": tr, "traffic_note": note,
                    "algorithmic_bytes_per_launch": s["bytes"] / max(s["launches"], 1),
                    "avg_launch_ms": s["ms"] / max(s["launches"], 1),
                    "launches_per_step": s["launches"] // a.steps}

        total_imgs = B * a.steps * world
        if a.fp16:
            compute_dtype = "f16"
        elif a.fp32_mfma:
            compute_dtype = "f32"
        else:
            # fp32 tensors and fp32 accumulation; products formed from fp16 (high, low) pairs on
            # the fp16 matrix instruction -- fp32-level accuracy, see precision_check below
            compute_dtype = "f32s"
        res = {
            "metric": "images/sec whole-node, 512x512 ctdet", "value": total_imgs / dt,
            "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": compute_dtype, "data": "stub" if stub else "synthetic",
            "config": {"workload": "%s %s %dx%d, batch %d per GPU (BASELINE configs[%d]%s), network "
                                   "+ fused sigmoid/peak-NMS/top-K decode, K=%d"
                                   % (a.task, a.arch, a.res, a.res, B, a.config,
                                      "" if standard else ", modified", opt.K),
                       "global_batch": B * world, "parallelism": "image-sharded x%d" % world,
                       "gflop_per_image": plan.flops / B / 1e9},
            "world_size_seen": world,
            "rccl_ranks_seen": world if (dist is not None and backend == "nccl") else None,
            "rank_devices": rank_devices,
            "backend": ("%s (RCCL over xGMI)" % backend if backend == "nccl" else backend) if dist is not None else None,
            "weight_broadcast_bytes": bcast_bytes, "weight_broadcast_ms": bcast_ms,
            "per_rank_img_s": per_rank_rate, "cross_rank_agreement": agreement,
            "box_calibration": calib,
            "headline_over_calibration": None if not calib else {
                "img_s_per_mfma_TFLOPs": total_imgs / dt / world / calib["mfma_f16_TFLOPs"],
                "img_s_per_copy_TBs": total_imgs / dt / world / calib["copy_TBs"]},
            "range_tracking_off_leg": track_leg,
            "clock_under_load": clock,
            "roofline": roof(dom, "mfma" if kinds[dom]["flops"] > 0 else "hbm"),
            "roofline_dcn_mfma": roof("dcn", "mfma") if "dcn" in kinds else None,
            "roofline_dcn_hbm": roof("dcn", "hbm") if "dcn" in kinds else None,
            "roofline_decode_hbm": roof("decode", "hbm"),
            "dominant_launch": None if stub else dominant_launch(
                det, images, plan, 5,
                F16_MFMA_PEAK_TF if a.fp16 else (F32_MFMA_PEAK_TF if a.fp32_mfma else F32S_MFMA_PEAK_TF),
                a.config),
            "pmc_profile": pmc_file,
            "pmc_profile_head": None if pmc_file is None else (pmc_stale[0] if pmc_stale else kernel_tree_sha()),
            "kernel_tree_head": kernel_tree_sha(),
            "pmc_profile_stale": bool(pmc_stale),
            "traffic_source": None if pmc_file is None else
            ("REFUSED: profiles/%s was taken on kernel sources %s, this run is %s -- traffic is null until the "
             "PMC passes are re-taken (tools/profile_round.sh)" % (pmc_file, pmc_stale[0], pmc_stale[1])) if pmc_stale else
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed as profiles/%s "
            "(counters cannot be read from inside the run)" % pmc_file,
            "precision_check": precision_check(dev) if not (a.fp16 or stub) else None,
            "fp32_mfma_leg": fp32_leg,
            "f32s_range": None if (a.fp16 or a.fp32_mfma) else {
                "clean": bool(range_clean),
                "calibrations": det.model.__dict__.get("_calibrations", 0),
                "policy": "per-tensor power-of-two exponents from one fp32 calibration pass; every "
                          "split site max-es |value| into range words, read inside the timed region"},
            "time_share": {k: round(v["ms"] / (dt * 1e3), 4) for k, v in kinds.items()},
        }
        if world == 1 and standard and a.config == 1 and not a.no_secondary and not stub:
            # the reference's own operating point -- ONE image per step (src/test.py:60-62, MODEL_ZOO: 7 ms on a
            # TITAN Xp for this network) -- on the headline's detector: det.run_batch on a batch of one, the same
            # entry point, no markers inside the loop
            try:
                one = synth.images(1, a.res, a.res, seed=7).to(dev)
                for _ in range(10):
                    det.run_batch(one)
                torch.cuda.synchronize()
                assert det.range_ok(one), "f32s range check failed at batch 1"
                n1 = 100
                t1 = time.perf_counter()
                for _ in range(n1):
                    det.run_batch(one)
                ok1 = det.range_ok()
                torch.cuda.synchronize()
                b1 = (time.perf_counter() - t1) / n1 * 1e3
                res["batch_1"] = {"b1_ms_per_step": b1, "img_s": 1e3 / b1, "steps": n1, "range_clean": bool(ok1),
                                  "what": "network + decode for ONE 512x512 image per step (device-resident input), "
                                          "launches per step %d" % (len(det.model.plan_for(1, a.res, a.res, dev).b.ops) + 1)}
            except Exception as e:
                res["batch_1"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            # BASELINE configs[2..4], a few steps each, behind the headline loop (never inside it)
            sec = {}
            for cid in (2, 3, 4):
                try:
                    sec["configs[%d]" % cid] = secondary_config(cid, a.res, a.secondary_steps, dev)
                except Exception as e:      # a secondary leg must never cost the headline line
                    sec["configs[%d]" % cid] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            res["secondary_configs"] = sec
        if world == 1 and not a.no_cpu_baseline and not stub:
            sd = {k: v.detach().cpu() for k, v in det.model.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(a.task, a.arch, sd, list(opt.heads), a.res,
                                               a.cpu_seconds)
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
