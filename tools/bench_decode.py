"""Decode kernel timing + phase ablation (GPU box only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import native
lib = native.lib()
dev = torch.device("cuda:0")
B, C, H, W, K = int(os.environ.get("B", 32)), 80, 128, 128, 100
g = torch.Generator().manual_seed(0)
logits = (2 * torch.randn((B, C, H, W), generator=g) - 2.19).to(dev)
wh = (40 * torch.rand((B, 2, H, W), generator=g)).to(dev)
reg = torch.rand((B, 2, H, W), generator=g).to(dev)
dets = torch.empty((B, K, 6), device=dev)
inds = torch.empty((B, K), device=dev, dtype=torch.int32)
n = lib.cn_ctdet_decode_workspace_bytes(B, C, H, W, K)
ws = torch.empty(n, device=dev, dtype=torch.uint8)
alg = B * (C * H * W * 4 + 4 * H * W * 4 + K * 24)


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


for name, flags in (("image-level select, logits in (default)", 1), ("image-level select, post-sigmoid in", 0),
                    ("per-band select of round 1 (flag 2048)", 1 | 2048)):
    ms = run(flags)
    print("%-44s %7.3f ms  %7.1f GB/s" % (name, ms, alg / ms / 1e6))
