"""Tensor helpers (mirror of src/lib/models/utils.py:28-50), kept on the device."""
import torch


def flip_tensor(x):
    return torch.flip(x, [3])


def flip_lr(x, flip_idx):
    # models/utils.py:33-39 -- the reference round-trips through NumPy on the host
    y = torch.flip(x, [3]).clone()
    for a, b in flip_idx:
        tmp = y[:, a].clone()
        y[:, a] = y[:, b]
        y[:, b] = tmp
    return y


def flip_lr_off(x, flip_idx):
    # models/utils.py:41-50
    y = torch.flip(x, [3])
    shape = y.shape
    y = y.reshape(shape[0], 17, 2, shape[2], shape[3]).clone()
    y[:, :, 0] *= -1
    for a, b in flip_idx:
        tmp = y[:, a].clone()
        y[:, a] = y[:, b]
        y[:, b] = tmp
    return y.reshape(shape)
