"""dla_34 multi_pose (BASELINE configs[3]): network + multi_pose_decode per batch."""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth
from centernet_amd.opts import opts
from centernet_amd.detectors.detector_factory import detector_factory
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with contextlib.redirect_stdout(sys.stderr):
    opt = opts().init(["multi_pose", "--arch", "dla_34"])
    det = detector_factory[opt.task](opt)
synth.fill_state_dict_(det.model, 317)
det.model.invalidate_plans()
dev = opt.device
x = synth.images(B, 512, 512, 0).to(dev)
from centernet_amd.decode import multi_pose_decode
def step():
    with torch.no_grad():
        o = det.model(x)[-1]
        return o, multi_pose_decode(o['hm'], o['wh'], o['hps'], reg=o['reg'], hm_hp=o['hm_hp'],
                                    hp_offset=o['hp_offset'], K=opt.K, apply_sigmoid=True)
for _ in range(3): step()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 10
tn = td = 0.0
for _ in range(n):
    e[0].record()
    with torch.no_grad():
        o = det.model(x)[-1]
    e[1].record()
    d = multi_pose_decode(o['hm'], o['wh'], o['hps'], reg=o['reg'], hm_hp=o['hm_hp'],
                          hp_offset=o['hp_offset'], K=opt.K, apply_sigmoid=True)
    e[2].record()
    torch.cuda.synchronize()
    tn += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
print("dla_34 multi_pose B=%d: net %.3f ms  decode %.3f ms  -> %.0f img/s" % (B, tn / n, td / n, B / ((tn + td) / n) * 1e3))
