"""Device-side flip helpers for flip-test averaging (semantics of src/lib/models/utils.py:28-50;
the reference round-trips through NumPy on the host, these stay on the device and express the
left/right joint swap as one index permutation)."""
import torch


def flip_tensor(x):
    """Mirror a (B, C, H, W) map along x."""
    return torch.flip(x, [3])


def _joint_permutation(n, flip_idx, device):
    perm = list(range(n))
    for a, b in flip_idx:
        perm[a], perm[b] = perm[b], perm[a]
    return torch.tensor(perm, device=device, dtype=torch.long)


def flip_lr(x, flip_idx):
    """Mirrored joint heat-maps: flip along x and exchange left / right joints (utils.py:33-39)."""
    return torch.flip(x, [3]).index_select(1, _joint_permutation(x.shape[1], flip_idx, x.device))


def flip_lr_off(x, flip_idx):
    """Mirrored joint offsets (B, 2J, H, W): flip along x, negate the x component, exchange
    left / right joints (utils.py:41-50)."""
    b, c2, h, w = x.shape
    joints = c2 // 2
    y = torch.flip(x, [3]).reshape(b, joints, 2, h, w)
    sign = torch.tensor([-1.0, 1.0], device=x.device, dtype=x.dtype).view(1, 1, 2, 1, 1)
    y = (y * sign).index_select(1, _joint_permutation(joints, flip_idx, x.device))
    return y.reshape(b, c2, h, w)
