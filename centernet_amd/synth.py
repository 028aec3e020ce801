"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

Everything is drawn from ``numpy.random.RandomState`` (a frozen legacy stream), so the
same seed gives the same bits here, on the GPU box and inside
``tests/golden/gen_golden.py`` -- independent of torch's initialisers.

Recipe: He-normal conv weights; BatchNorm made non-trivial
(gamma ~ U(0.75,1.25) -- U(0.15,0.35) on the closing BN of a residual branch so the trunk
stays O(1) -- beta, running_mean ~ N(0,0.1), running_var ~ U(0.8,1.25));
``conv_offset_mask`` -- zero in the reference (DCNv2/dcn_v2.py:60-62) -- gets weights so
that offsets spread over a few pixels and masks vary, otherwise the deformable gather
would never leave the regular grid; ``hm`` output bias = -2.19 (resnet_dcn.py:165-166).
"""
import math

import numpy as np
import torch


def fill_state_dict_(module, seed=317):
    """Overwrite every parameter / buffer of ``module`` in place; returns the module."""
    import zlib
    sd = module.state_dict()
    keys = sorted(sd.keys())
    new = {}
    for k in keys:
        v = sd[k]
        if k.endswith("num_batches_tracked"):
            continue
        # one stream per tensor, keyed by (seed, name): independent of registration order
        # and of which other tensors exist (e.g. the unused base.fc of DLA checkpoints)
        rng = np.random.RandomState((seed * 1000003 + zlib.crc32(k.encode())) & 0x7FFFFFFF)
        shape = tuple(v.shape)
        prefix = k.rsplit(".", 1)[0] if "." in k else ""
        leaf = k.rsplit(".", 1)[-1]
        is_bn = (prefix + ".running_mean") in sd if prefix else ("running_mean" in sd)
        if leaf == "running_mean":
            a = rng.standard_normal(shape) * 0.1
        elif leaf == "running_var":
            a = rng.uniform(0.8, 1.25, size=shape)
        elif is_bn and leaf == "weight":
            # the last BN of a residual branch is damped (as zero-init-residual training
            # leaves it) so that activations stay O(1) through the trunk: with O(100)
            # activations the offset convolution would throw every deformable sample
            # hundreds of pixels outside the map and the network becomes chaotic in its
            # inputs, which no real checkpoint is.
            tail = prefix.rsplit(".", 1)[-1]
            if tail in ("bn2", "bn3"):
                a = rng.uniform(0.15, 0.35, size=shape)
            elif k.startswith(("cnvs.", "cnvs_.", "inters_.")):
                # hourglass: each stack multiplies the activation scale by ~6 (identity
                # skips + five merge adds); the BN that closes a stack brings it back to O(1)
                a = rng.uniform(0.1, 0.2, size=shape)
            elif ".actf." in k:
                # BN after a DCN whose input is a sum of two maps (IDAUp node): keep the
                # up-sampling pyramid from doubling its variance at every node
                a = rng.uniform(0.45, 0.75, size=shape)
            else:
                a = rng.uniform(0.75, 1.25, size=shape)
        elif is_bn and leaf == "bias":
            a = rng.standard_normal(shape) * 0.1
        elif v.dim() == 4:
            if "conv_offset_mask" in k:
                fan_in = shape[1] * shape[2] * shape[3]
                a = rng.standard_normal(shape) * (0.3 / math.sqrt(fan_in))
            elif shape[1] == 1 and (".up_" in k or _is_transposed(module, prefix)):
                # depthwise up-sampling (IDAUp): bilinear kernel (what fill_up_weights,
                # pose_dla_dcn.py:329-338, initialises and training keeps close to) x jitter
                kk = shape[2]
                f = int(math.ceil(kk / 2))
                c = (2 * f - 1 - f % 2) / (2.0 * f)
                g = np.array([1 - abs(i / f - c) for i in range(kk)])
                a = (g[:, None] * g[None, :])[None, None] * rng.uniform(0.8, 1.2, size=shape)
            elif ".up_" in k or _is_transposed(module, prefix):
                # ConvTranspose2d weight is (Cin, Cout/groups, kh, kw)
                fan_in = max(1, shape[0] * shape[2] * shape[3] // 4)
                a = rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)
            else:
                fan_in = shape[1] * shape[2] * shape[3]
                gain = 1.6 if _is_dcn(module, prefix) else 1.0  # mask ~0.5 halves the signal
                if _is_head_out(k, sd):
                    # logits std ~1 around the -2.19 bias for every arch (no saturated, tied
                    # scores): scale by the head's input width F and hidden width hc
                    first = sd.get(k.split(".")[0] + ".0.weight")
                    F_in = first.shape[1] if first is not None and first.dim() == 4 else shape[1]
                    gain = 0.5 * (64.0 / F_in) ** 0.66 * (shape[1] / 64.0) ** 0.8
                a = rng.standard_normal(shape) * (gain * math.sqrt(2.0 / fan_in))
        elif leaf == "bias":
            if "conv_offset_mask" in k:
                a = rng.standard_normal(shape)
            elif _is_hm_out(k, sd):
                a = np.full(shape, -2.19)
            else:
                a = rng.standard_normal(shape) * 0.1
        else:
            a = rng.standard_normal(shape) * 0.1
        new[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(shape)
    with torch.no_grad():
        for k, t in new.items():
            sd[k].copy_(t)
    if hasattr(module, "invalidate_plans"):
        module.invalidate_plans()
    return module


def _is_dcn(module, prefix):
    try:
        m = module.get_submodule(prefix) if prefix else module
    except Exception:
        return False
    return hasattr(m, "conv_offset_mask")


def _is_transposed(module, prefix):
    try:
        m = module.get_submodule(prefix) if prefix else module
    except Exception:
        return False
    return isinstance(m, torch.nn.ConvTranspose2d)


_TRUNK = ("conv", "bn", "layer", "deconv", "base", "dla_up", "ida_up", "pre", "kps", "cnvs",
          "inters", "cnvs_", "inters_")


def _is_head_out(key, sd):
    """weight/bias of the LAST conv of a head module ('hm.2.weight', 'wh.bias', ...)."""
    parts = key.split(".")
    if len(parts) < 2 or parts[0].startswith(_TRUNK):
        return False
    return _is_last_of_head(parts, sd)


def _is_hm_out(key, sd):
    # last conv of a head whose name contains 'hm' (hm, hm_hp): 'hm.2.bias' or 'hm.bias'
    parts = key.split(".")
    if "hm" not in parts[0]:
        return False
    return _is_last_of_head(parts, sd)


def _is_last_of_head(parts, sd):
    """True if the key belongs to the LAST conv of its head: 'hm.weight', 'hm.2.weight'
    (resnet/dla heads), 'hm.1.1.weight' (hourglass: head[stack][1])."""
    path = parts[:-1]
    if len(path) == 0:
        return False
    if len(path) == 1:
        return True
    if not path[-1].isdigit():
        return False
    prefix = ".".join(path[:-1]) + "."
    idx = int(path[-1])
    for k in sd:
        if k.startswith(prefix):
            rest = k[len(prefix):].split(".")
            if rest[0].isdigit() and int(rest[0]) > idx:
                return False
    return True


def images(B, H=512, W=512, seed=0):
    """Already-normalised synthetic input batch, (B,3,H,W) fp32, CPU."""
    rng = np.random.RandomState(seed)
    return torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32))


def heatmap(shape, seed=0):
    """Post-sigmoid-like scores in (0,1) built with exact arithmetic only: u^2 * v^2.
    Mostly small values with a sparse upper tail, so the top-K scores are well separated
    (no float32 ties among them) the way a trained heat-map's peaks are."""
    rng = np.random.RandomState(seed)
    u = rng.random_sample(shape).astype(np.float32)
    v = rng.random_sample(shape).astype(np.float32)
    return (u * u) * (v * v)


def uniform(shape, lo, hi, seed):
    rng = np.random.RandomState(seed)
    return (rng.random_sample(shape) * (hi - lo) + lo).astype(np.float32)


def normal(shape, std, seed, mean=0.0):
    rng = np.random.RandomState(seed)
    return (rng.standard_normal(shape) * std + mean).astype(np.float32)


def rescale_activations_(module, log2_scale):
    """Multiply every INTERNAL activation of a ResNet-type network (``networks.resnet.PoseResNet``)
    by 2^log2_scale while leaving its outputs unchanged: the first BatchNorm scales its
    (gamma, beta), every later one its (running_mean, beta); DCN offset convolutions divide
    their weights (offsets and masks stay what they were) and DCN / first-head-conv biases
    scale; the last convolution of each head divides its weights.  Powers of two commute with
    fp32 rounding, so the rescaled network is the SAME function bit for bit in fp32 -- a network
    whose feature maps sit at 1e-4 or 1e+4 instead of O(1), for the f32s range tests."""
    s = float(2.0 ** log2_scale)
    with torch.no_grad():
        first_bn = True
        for name, m in module.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                if first_bn:
                    assert name == "bn1", name
                    m.weight.mul_(s)
                    first_bn = False
                else:
                    m.running_mean.mul_(s)
                m.bias.mul_(s)
            elif hasattr(m, "conv_offset_mask"):
                m.conv_offset_mask.weight.div_(s)
                m.bias.mul_(s)
        for h in module.heads:
            seq = getattr(module, h)
            assert isinstance(seq, torch.nn.Sequential)
            seq[0].bias.mul_(s)
            seq[-1].weight.div_(s)
    if hasattr(module, "invalidate_plans"):
        module.invalidate_plans()
    return module
