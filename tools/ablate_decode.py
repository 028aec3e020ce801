"""Stage ablation of the one-launch decode (GPU box only; needs a -DCN_ABLATE_DECODE variant build:
tools/build_variant.sh dabl -DCN_ABLATE_DECODE, CENTERNET_AMD_LIB=centernet_amd/variants/libcenternet_amd_dabl.so).
The kernel leaves after stage s = (flags >> 16) & 15: 1 loads, 2 keys, 3 threshold, 4 list, 5 plane select +
hand-on, 6 arrival, 0 everything."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native
lib = native.lib()
dev = torch.device("cuda:0")
B, C, H, W, K = int(os.environ.get("B", 32)), 80, 128, 128, 100
g = torch.Generator().manual_seed(0)
wh = (40 * torch.rand((B, 2, H, W), generator=g)).to(dev)
reg = torch.rand((B, 2, H, W), generator=g).to(dev)
kind = os.environ.get("HEAT", "iid")
if kind == "iid":
    logits = (2 * torch.randn((B, C, H, W), generator=g) - 2.19).to(dev)
else:
    lo = torch.nn.functional.interpolate(torch.randn((B, C, H // 4, W // 4), generator=g), scale_factor=4,
                                         mode="bilinear", align_corners=False)
    logits = (-6.9 + 0.4 * lo + 0.05 * torch.randn((B, C, H, W), generator=g)).to(dev)
dets = torch.empty((B, K, 6), device=dev)
inds = torch.empty((B, K), device=dev, dtype=torch.int32)
n = lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K)
ws = torch.zeros(n, device=dev, dtype=torch.uint8)


def run(flags, iters=30):
    def call():
        rc = lib.cn_ctdet_decode_f32(native.ptr(logits), native.ptr(wh), native.ptr(reg), B, C, H, W, K, 0,
                                     flags, native.ptr(dets), native.ptr(inds), native.ptr(ws), n,
                                     native.stream_ptr())
        assert rc == 0, rc
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        call()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


print("lib:", native.LIB_PATH, "heat:", kind)
for sig in (1,):
    for st, name in ((1, "loads"), (2, "+keys"), (3, "+threshold"), (4, "+list"), (5, "+plane select / hand-on"),
                     (6, "+arrival"), (0, "everything"), (0 | 8, "everything, image-major")):
        ms = min(run(sig | ((st & 7) << 16)) for _ in range(3))
        print("sigmoid=%d  stage %d %-26s %7.3f ms" % (sig, st, name, ms))
