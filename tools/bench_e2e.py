"""Host-boundary (PCIe-inclusive) rates of the task API: uint8 frames in host memory ->
result dicts in host memory.  python tools/bench_e2e.py [arch]"""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centernet_amd import synth
from centernet_amd.opts import opts
from centernet_amd.detectors.detector_factory import detector_factory
arch = sys.argv[1] if len(sys.argv) > 1 else "resdcn_18"
with contextlib.redirect_stdout(sys.stderr):
    opt = opts().init(["ctdet", "--arch", arch])
    det = detector_factory[opt.task](opt)
synth.fill_state_dict_(det.model, 317)
det.model.invalidate_plans()
rng = np.random.RandomState(0)
frames = [rng.randint(0, 256, (512, 512, 3)).astype(np.uint8) for _ in range(32)]
for host_pre in (True, False):
    opt.host_pre_process = host_pre
    for f in frames[:3]: det.run(f)
    keys = ("tot", "load", "pre", "net", "dec", "post", "merge")
    acc = dict.fromkeys(keys, 0.0)
    t = time.perf_counter()
    for f in frames:
        r = det.run(f)
        for k in keys: acc[k] += r[k]
    dt = time.perf_counter() - t
    print("%s run(frame) %-11s %.2f ms/img (%.0f img/s)  " % (arch, "host-pre" if host_pre else "device-pre", dt / 32 * 1e3, 32 / dt)
          + " ".join("%s %.2f" % (k, acc[k] / 32 * 1e3) for k in keys))
for B in (8, 32):
    fr = frames[:B]
    det.run_frames(fr); det.run_frames(fr)
    t = time.perf_counter()
    n = 5
    for _ in range(n): det.run_frames(fr)
    dt = (time.perf_counter() - t) / n
    print("%s run_frames(B=%d) %.2f ms/batch  %.0f img/s (uint8 H2D + device pre-process + net + decode + D2H + host post)" % (arch, B, dt * 1e3, B / dt))

# pipelined form: batches staged / uploaded / collected around the device work (run_frames_stream)
for B in (8, 32):
    nb = 24
    batches = [[frames[(i + j) % 32] for j in range(B)] for i in range(nb)]
    for _ in det.run_frames_stream(iter(batches[:4])): pass
    t = time.perf_counter()
    n = sum(len(r) for r in det.run_frames_stream(iter(batches)))
    dt = time.perf_counter() - t
    print("%s run_frames_stream(B=%d) %.2f ms/batch  %.0f img/s (pinned staging by 4 threads + async uint8 H2D on a copy "
          "stream + batched device pre-process + net + decode + device tail + D2H, pipelined 3 deep, %d batches)"
          % (arch, B, dt / nb * 1e3, n / dt, nb))
