#!/usr/bin/env python
"""Headline benchmark: images/s of the CenterNet ctdet hot path (network + decode) on
512x512 synthetic input, ResNet-18-DCN, batch 32 per GPU, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = detector.run_batch(images): backbone + DCN + heads + fused sigmoid/decode over
one device-resident batch.  Images shard over ranks (weak scaling, no collective in the
loop); rank 0 broadcasts the flat weight buffer once at start-up over RCCL.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md)


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC summary (tools/pmc_traffic.py,
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); summed per kernel class."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        raw = json.load(f)
    cls = {"conv": 0.0, "dcn": 0.0, "decode": 0.0, "maxpool": 0.0}
    n = {"conv": 0, "dcn": 0, "decode": 0, "maxpool": 0}
    steps = max(1, raw.get("nms_topk_kernel", {}).get("launches", 1))
    for k, v in raw.items():
        if k.startswith("igemm_kernel"):
            c = "dcn" if k.rstrip(">").split(",")[5].strip() == "2" else "conv"
        elif k.startswith(("stem_conv", "stem_persist", "splitk_reduce", "conv3x3s1_kernel", "conv16_kernel")):
            c = "conv"
        elif k.startswith(("nms_topk", "merge_topk")):
            c = "decode"
        elif k.startswith("maxpool"):
            c = "maxpool"
        else:
            continue
        cls[c] += v["hbm_bytes_per_launch"] * v["launches"] / steps   # bytes per forward step
        n[c] += v["launches"] // steps
    return {c: (cls[c], n[c]) for c in cls if n[c]}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--tune", action="append", default=[],
                   help="KEY=VALUE for cn_set_tuning (A/B experiments; not used by the driver)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    p.add_argument("--arch", default="resdcn_18")
    p.add_argument("--res", type=int, default=512)
    p.add_argument("--fp16", action="store_true",
                   help="fp16 activations/weights, fp32 accumulate (configs[4], hourglass only)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-images", type=int, default=16)
    p.add_argument("--per-op", action="store_true", help="print per-launch timings to stderr")
    return p.parse_args()


def cpu_baseline(arch, state_dict, heads, res, n_images):
    """The CPU restatement of the same path (oracle/: torch-CPU dense ops, i.e. the
    reference's own arithmetic, + C DCNv2 + C decode) timed on the host cores."""
    from centernet_amd import synth
    from oracle import net_oracle, cref
    cref.lib()
    cores = min(os.cpu_count() or 1, 64)   # more threads than this slows batch-1 convs down
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    x = synth.images(1, res, res, seed=1)
    net_oracle.ctdet_process(arch, state_dict, x, heads)  # warm-up
    t0 = time.time()
    done = 0
    while done < n_images and (time.time() - t0) < 20.0:   # bounded sample: <= ~20 s of CPU work
        net_oracle.ctdet_process(arch, state_dict, synth.images(1, res, res, seed=2 + done), heads)
        done += 1
    dt = time.time() - t0
    return {"value": done / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%d images %dx%d batch 1, oracle/net_oracle.ctdet_process (torch-CPU convs "
                      "+ C DCNv2 + C decode), %.1f s" % (done, res, res, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # BENCH_DEVICE / BENCH_BACKEND exist only so the multi-rank control flow can be exercised on
    # a single-GPU box (two ranks sharing cuda:0 over gloo); the driver never sets them.
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from centernet_amd import synth, native
    for kv in a.tune:
        k, v = kv.split("=")
        assert native.lib().cn_set_tuning(int(k), int(v)) == 0, kv
    from centernet_amd.opts import opts
    from centernet_amd.detectors import detector_factory
    from centernet_amd.sharding import broadcast_weights

    opt = opts().init(["ctdet", "--arch", a.arch, "--input_res", str(a.res)])
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the detector prints 'Creating model...'
        det = detector_factory[opt.task](opt)
    if rank == 0:
        synth.fill_state_dict_(det.model, 317)
    if world > 1:
        broadcast_weights(det.model, src=0)      # one flat RCCL broadcast, then no collectives
    det.model.invalidate_plans()
    if a.fp16:
        det.model.half_compute()
    B = a.batch
    images = synth.images(B, a.res, a.res, seed=100 + rank).to(dev)

    for _ in range(max(a.warmup, 1)):
        dets = det.run_batch(images)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    plan = det.model.plan_for(B, a.res, a.res, dev)
    # HIP events on the launch stream, inside the timed region.  A marker after every launch
    # costs ~7 us of bubble each (35 per step), so by default events sit only where the kernel
    # class changes (the per-class sums stay exact); --per-op records one after every launch.
    metas = plan.b.meta
    nops = len(metas)
    if a.per_op:
        bounds = list(range(nops))
    else:
        bounds = [i for i in range(nops) if i == nops - 1 or metas[i]["kind"] != metas[i + 1]["kind"]]
    bset = set(bounds)
    seg_first = [0] + [b + 1 for b in bounds[:-1]]   # first op of every segment
    ev_all = []
    e_dec0, e_dec1 = [], []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        evs = []
        with torch.no_grad():
            out = plan.run(images, events=evs, event_after=bset)
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            from centernet_amd.decode import ctdet_decode
            dets = ctdet_decode(out["hm"], out["wh"], reg=out["reg"], K=opt.K, apply_sigmoid=True)
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
        ev_all.append(evs); e_dec0.append(e0); e_dec1.append(e1)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # ---- per-kernel-class time from the HIP events recorded inside the timed region
        kinds = {}
        for evs in ev_all:
            for si, (first, last) in enumerate(zip(seg_first, bounds)):
                ms = evs[si].elapsed_time(evs[si + 1])
                k = kinds.setdefault(metas[first]["kind"], {"ms": 0.0, "flops": 0, "bytes": 0, "launches": 0})
                k["ms"] += ms
                for m in metas[first:last + 1]:
                    k["flops"] += m["flops"]; k["bytes"] += m["bytes"]; k["launches"] += 1
        if a.per_op:
            for i, m in enumerate(plan.b.meta):
                ms = sum(evs[i].elapsed_time(evs[i + 1]) for evs in ev_all) / a.steps
                kind, act = plan.b.trace[i]
                print("op %2d %-7s out(B,H,W,C)=(%d,%d,%d,%d) %8.3f ms  %7.2f TF/s  %7.1f GB/s" % (
                    i, kind, act.B, act.H, act.W, act.C, ms, m["flops"] / ms / 1e9 if ms else 0,
                    m["bytes"] / ms / 1e6 if ms else 0), file=sys.stderr)
        dec_ms = sum(s.elapsed_time(e) for s, e in zip(e_dec0, e_dec1))
        C, Ho = opt.num_classes, a.res // 4
        dec_bytes = B * (C * Ho * Ho * 4 + 2 * 2 * Ho * Ho * 4 + opt.K * 6 * 4)
        kinds["decode"] = {"ms": dec_ms, "flops": 0, "bytes": dec_bytes * a.steps,
                           "launches": 2 * a.steps}
        dom = max(kinds, key=lambda k: kinds[k]["ms"])
        pmc = pmc_traffic() if (a.arch == "resdcn_18" and B == 32 and a.res == 512 and not a.fp16) else {}

        def traffic(k):
            # measured HBM bytes per launch of this kernel class (committed PMC profile of the
            # same command), or null when the profile does not cover this configuration
            if k not in pmc:
                return None
            byts, launches = pmc[k]
            return byts / max(launches, 1)

        def mfma_roof(k):
            s = kinds[k]
            ach = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 else 0.0
            peak = F16_MFMA_PEAK_TF if a.fp16 else F32_MFMA_PEAK_TF
            return {"kernel": k, "bound": "mfma", "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic(k),
                    "avg_launch_ms": s["ms"] / max(s["launches"], 1), "launches_per_step": s["launches"] // a.steps}

        def hbm_roof(k):
            s = kinds[k]
            ach = s["bytes"] / (s["ms"] * 1e-3) / 1e9 if s["ms"] > 0 else 0.0
            return {"kernel": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic(k),
                    "avg_launch_ms": s["ms"] / max(s["launches"], 1), "launches_per_step": s["launches"] // a.steps}

        roofline = mfma_roof(dom) if kinds[dom]["flops"] > 0 else hbm_roof(dom)
        total_imgs = B * a.steps * world
        res = {
            "metric": "images/sec whole-node, 512x512 ctdet", "value": total_imgs / dt,
            "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if a.fp16 else "f32", "data": "synthetic",
            "config": {"workload": "ctdet %s %dx%d, batch %d per GPU (BASELINE configs[1]), "
                                   "network + fused sigmoid/peak-NMS/top-K decode, K=%d"
                                   % (a.arch, a.res, a.res, B, opt.K),
                       "global_batch": B * world, "parallelism": "image-sharded x%d" % world,
                       "gflop_per_image": plan.flops / B / 1e9},
            "roofline": roofline,
            "roofline_dcn_mfma": mfma_roof("dcn") if "dcn" in kinds else None,
            "roofline_dcn_hbm": hbm_roof("dcn") if "dcn" in kinds else None,
            "roofline_decode_hbm": hbm_roof("decode"),
            "time_share": {k: round(v["ms"] / (dt * 1e3), 4) for k, v in kinds.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            sd = {k: v.detach().cpu() for k, v in det.model.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(a.arch, sd, list(opt.heads), a.res, a.cpu_images)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
