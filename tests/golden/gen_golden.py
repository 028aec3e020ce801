#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the development container only (it needs /root/reference, which does not exist
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What is executed from the reference (imported, never copied):
  * src/lib/models/decode.py : _nms, _topk, _topk_channel, ctdet_decode,
    multi_pose_decode -- on CPU tensors (torch 2.10).
  * src/lib/models/networks/msra_resnet.py : PoseResNet(BasicBlock,[2,2,2,2]) built
    directly (the get_pose_net factory downloads ImageNet weights) -> 'res_18'.
  * src/lib/models/networks/resnet_dcn.py : PoseResNet built directly with the
    unbuildable ``DCNv2._ext`` stubbed and ``DCN.forward`` routed to
    oracle/dcn_v2_oracle.c (the reference's DCN has no runnable implementation here) ->
    'resdcn_18'.  This pins the module GRAPH and every dense layer; the DCN arithmetic
    itself is pinned separately (tests/test_oracle_dcn.py).

Inputs are regenerated from seeds (centernet_amd.synth, numpy RandomState) by the tests,
so only outputs + an input checksum are stored.
"""
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/lib"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from centernet_amd import synth  # noqa: E402
from oracle import cref  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        if a is not None:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------- decode
DECODE_CASES = {
    # name: (B, C, H, W, K, reg, cat_spec_wh)
    "ctdet_coco": (2, 80, 128, 128, 100, True, False),
    "ctdet_small_catspec": (1, 3, 16, 16, 10, False, True),
    "ctdet_rect": (3, 5, 24, 40, 20, True, False),
    "ctdet_odd": (2, 4, 17, 23, 7, True, False),
}


def decode_inputs(name):
    B, C, H, W, K, use_reg, cat_spec = DECODE_CASES[name]
    seed = sum(map(ord, name))
    heat = synth.heatmap((B, C, H, W), seed)
    wh = synth.uniform((B, 2 * C if cat_spec else 2, H, W), 0.0, 40.0, seed + 1)
    reg = synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 2) if use_reg else None
    return heat, wh, reg, K, cat_spec


POSE_CASES = {
    # name: (B, H, W, K, reg, hm_hp, hp_offset)
    "pose_full": (2, 128, 128, 100, True, True, True),
    "pose_no_hm_hp": (1, 32, 32, 20, True, False, False),
    "pose_no_offsets": (2, 24, 40, 30, False, True, False),
}


def pose_inputs(name):
    B, H, W, K, use_reg, use_hm_hp, use_off = POSE_CASES[name]
    seed = 1000 + sum(map(ord, name))
    J = 17
    heat = synth.heatmap((B, 1, H, W), seed)
    wh = synth.uniform((B, 2, H, W), 0.0, 60.0, seed + 1)
    kps = synth.normal((B, 2 * J, H, W), 8.0, seed + 2)
    reg = synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 3) if use_reg else None
    # keypoint heat-map: mostly small, a few strong peaks (so the 0.1 threshold bites)
    hm_hp = None
    if use_hm_hp:
        u = synth.heatmap((B, J, H, W), seed + 4)
        hm_hp = np.sqrt(np.sqrt(u)).astype(np.float32) * synth.heatmap((B, J, H, W), seed + 5)
        hm_hp = np.ascontiguousarray(hm_hp, dtype=np.float32)
    hp_offset = synth.uniform((B, 2, H, W), 0.0, 1.0, seed + 6) if use_off else None
    return heat, wh, kps, reg, hm_hp, hp_offset, K


def t(a):
    return None if a is None else torch.from_numpy(a.copy())


def gen_decode(ref_decode):
    out = {}
    meta = {}
    for name in DECODE_CASES:
        heat, wh, reg, K, cat_spec = decode_inputs(name)
        with torch.no_grad():
            dets = ref_decode.ctdet_decode(t(heat), t(wh), reg=t(reg), cat_spec_wh=cat_spec, K=K)
            nmsd = ref_decode._nms(t(heat))
            s, i, c, y, x = ref_decode._topk(nmsd, K=K)
            cs, ci, cy, cx = ref_decode._topk_channel(nmsd, K=K)
        out[name + "/dets"] = dets.numpy()
        out[name + "/topk_score"] = s.numpy()
        out[name + "/topk_inds"] = i.numpy().astype(np.int64)
        out[name + "/topk_clses"] = c.numpy().astype(np.int32)
        out[name + "/topk_ys"] = y.numpy()
        out[name + "/topk_xs"] = x.numpy()
        out[name + "/chan_score"] = cs.numpy()
        out[name + "/chan_inds"] = ci.numpy().astype(np.int64)
        if heat.size <= 4096:
            out[name + "/nms"] = nmsd.numpy()
        meta[name] = {"sha": sha(heat, wh, reg),
                      "min_gap": float(np.min(-np.diff(s.numpy(), axis=1)))}
    for name in POSE_CASES:
        heat, wh, kps, reg, hm_hp, hp_offset, K = pose_inputs(name)
        with torch.no_grad():
            dets = ref_decode.multi_pose_decode(t(heat), t(wh), t(kps), reg=t(reg), hm_hp=t(hm_hp),
                                                hp_offset=t(hp_offset), K=K)
        out[name + "/dets"] = dets.numpy()
        meta[name] = {"sha": sha(heat, wh, kps, reg, hm_hp, hp_offset)}
    np.savez_compressed(os.path.join(HERE, "decode_golden.npz"), **out)
    with open(os.path.join(HERE, "decode_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("decode goldens:", sorted(meta))


# --------------------------------------------------------------------------- networks
NET_HEADS = {"hm": 80, "wh": 2, "reg": 2}
POSE_HEADS = {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}
NET_INPUT = (1, 128, 128)  # B, H, W  (small so the fixture stays small)
NET_SEED = 317             # reference default seed, src/lib/opts.py:43-44


def stub_dcn_ext():
    """The reference's DCNv2 extension cannot be built (THC / torch.utils.ffi / nvcc are
    gone).  Make ``from ._ext import dcn_v2`` importable and route DCN.forward to the C
    oracle."""
    pkg = "models.networks.DCNv2._ext"
    m = types.ModuleType(pkg)
    m.dcn_v2 = types.ModuleType(pkg + ".dcn_v2")
    m.__path__ = []
    sys.modules[pkg] = m
    sys.modules[pkg + ".dcn_v2"] = m.dcn_v2
    from models.networks.DCNv2 import dcn_v2 as ref_dcn  # noqa: E402

    def forward(self, input):
        # same statements as DCNv2/dcn_v2.py:64-68, then the oracle instead of the CUDA op
        out = self.conv_offset_mask(input)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        y = cref.dcn_v2_forward(input.detach().numpy(), offset.detach().numpy(),
                                mask.detach().numpy(), self.weight.detach().numpy(),
                                self.bias.detach().numpy(), self.stride, self.padding,
                                self.dilation, self.deformable_groups)
        return torch.from_numpy(y)

    ref_dcn.DCN.forward = forward
    return ref_dcn


def gen_nets():
    from models.networks import msra_resnet
    stub_dcn_ext()
    from models.networks import resnet_dcn
    B, H, W = NET_INPUT
    x = synth.images(B, H, W, seed=0)
    out = {}
    meta = {"input_sha": sha(x.numpy())}
    for arch, mod in (("res_18", msra_resnet), ("resdcn_18", resnet_dcn), ("resdcn_101", resnet_dcn)):
        torch.manual_seed(NET_SEED)
        block, layers = mod.resnet_spec[int(arch.split("_")[1])]
        net = mod.PoseResNet(block, layers, dict(NET_HEADS), head_conv=64)
        synth.fill_state_dict_(net, NET_SEED)
        net.eval()
        with torch.no_grad():
            ret = net(x)[-1]
        for h in NET_HEADS:
            out["%s/%s" % (arch, h)] = ret[h].numpy()
        sd = net.state_dict()
        meta[arch] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                      "weights_sha": sha(*[sd[k].numpy() for k in sorted(sd)
                                           if not k.endswith("num_batches_tracked")])}
    # DLA-34 (pose_dla_dcn.DLASeg built directly, pretrained=False: no download), ctdet heads
    # and multi_pose heads (opts.py:321-330), head_conv 256 (opts.py:246)
    from models.networks import pose_dla_dcn
    for arch, heads in (("dla_34", NET_HEADS), ("dla_34_pose", POSE_HEADS)):
        net = pose_dla_dcn.DLASeg("dla34", dict(heads), pretrained=False, down_ratio=4,
                                  final_kernel=1, last_level=5, head_conv=256)
        synth.fill_state_dict_(net, NET_SEED)
        net.eval()
        with torch.no_grad():
            ret = net(x)[-1]
        for h in heads:
            out["%s/%s" % (arch, h)] = ret[h].numpy()
        sd = net.state_dict()
        meta[arch] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                      "weights_sha": sha(*[sd[k].numpy() for k in sorted(sd)
                                           if not k.endswith("num_batches_tracked")])}
    # Hourglass-104 (large_hourglass.HourglassNet, 2 stacks); the detector uses outs[-1]
    from models.networks import large_hourglass
    net = large_hourglass.HourglassNet(dict(NET_HEADS), 2)
    synth.fill_state_dict_(net, NET_SEED)
    net.eval()
    with torch.no_grad():
        ret = net(x)[-1]
    for h in NET_HEADS:
        out["hourglass/%s" % h] = ret[h].numpy()
    sd = net.state_dict()
    meta["hourglass"] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                         "weights_sha": sha(*[sd[k].numpy() for k in sorted(sd)
                                              if not k.endswith("num_batches_tracked")])}
    np.savez_compressed(os.path.join(HERE, "net_golden.npz"), **out)
    with open(os.path.join(HERE, "net_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("net goldens:", [k for k in meta if k != "input_sha"])


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not found at %s: goldens can only be regenerated in the "
                         "development container" % REF)
    sys.path.insert(0, REF)
    from models import decode as ref_decode
    gen_decode(ref_decode)
    gen_nets()


if __name__ == "__main__":
    main()
