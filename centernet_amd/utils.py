"""Flip-test averaging on the device (semantics of src/lib/models/utils.py:28-50 and of the
``(a + flip(b)) / 2`` lines of detectors/ctdet.py:34-37, detectors/multi_pose.py:44-55): the reference
round-trips through NumPy on the host; here ONE launch per map pair (``cn_flip_average_f32``) mirrors
image 1, exchanges left / right joints, negates the x components of joint offsets, optionally applies
the logistic in place, and averages."""
import torch

from . import native


def _tables(C, flip_idx, offsets, device):
    """(source channel per output channel, sign per output channel) of flip_lr / flip_lr_off."""
    if flip_idx is None:
        return None, None
    joints = C // 2 if offsets else C
    perm = list(range(joints))
    for a, b in flip_idx:
        perm[a], perm[b] = perm[b], perm[a]
    if offsets:      # (B, 2J, H, W): channel 2j = x, 2j + 1 = y of joint j
        src = [2 * perm[c // 2] + (c & 1) for c in range(C)]
        sign = [-1.0 if (c & 1) == 0 else 1.0 for c in range(C)]
        return (torch.tensor(src, device=device, dtype=torch.int32),
                torch.tensor(sign, device=device, dtype=torch.float32))
    return torch.tensor(perm, device=device, dtype=torch.int32), None


def flip_average(pair, flip_idx=None, offsets=False, sigmoid=False):
    """``pair`` (2, C, H, W) fp32 on the device -> (1, C, H, W): the mean of image 0 and the
    un-mirrored image 1.  ``flip_idx``: left / right joint pairs (utils.py:33-39); ``offsets``: the
    channels are (x, y) joint offsets whose x changes sign (utils.py:41-50); ``sigmoid``: the logistic
    is applied to both images first, in place (``hm.sigmoid_()``)."""
    if pair.dim() != 4 or pair.shape[0] != 2 or pair.dtype != torch.float32 or not pair.is_cuda \
            or not pair.is_contiguous():
        raise ValueError("flip_average needs a contiguous (2, C, H, W) fp32 HIP tensor")
    _, C, H, W = pair.shape
    key = (C, None if flip_idx is None else tuple(map(tuple, flip_idx)), bool(offsets), str(pair.device))
    tabs = _TABLES.get(key)
    if tabs is None:
        tabs = _TABLES[key] = _tables(C, flip_idx, offsets, pair.device)
    src, sign = tabs
    out = torch.empty((1, C, H, W), device=pair.device, dtype=torch.float32)
    native.check(native.lib().cn_flip_average_f32(native.ptr(pair), native.ptr(out), C, H, W, native.ptr(src),
                                                  native.ptr(sign), int(bool(sigmoid)), native.stream_ptr()),
                 "cn_flip_average_f32")
    return out


_TABLES = {}
