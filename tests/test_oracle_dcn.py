"""oracle/dcn_v2_oracle.c pinned by the reference's one known-answer test
(DCNv2/test.py:32-65) and by analytic cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cref


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def test_zero_offset_identity_reference_kat():
    """DCNv2/test.py:32-65: offsets 0, mask 0.5, centre-tap identity weight => 2*y == x."""
    N, inC, inH, inW = 2, 2, 4, 4
    x = _rand((N, inC, inH, inW), 0)
    offset = np.zeros((N, 18, inH, inW), np.float32)
    mask = np.full((N, 9, inH, inW), 0.5, np.float32)
    w = np.zeros((inC, inC, 3, 3), np.float32)
    for c in range(inC):
        w[c, c, 1, 1] = 1.0
    y = cref.dcn_v2_forward(x, offset, mask, w, np.zeros(inC, np.float32))
    assert np.abs(2 * y - x).max() < 1e-10


@pytest.mark.parametrize("acc_mode", [0, 1])
def test_zero_offset_unit_mask_is_conv2d(acc_mode):
    x = _rand((2, 8, 9, 11), 1)
    w = _rand((6, 8, 3, 3), 2, 0.2)
    b = _rand((6,), 3)
    offset = np.zeros((2, 18, 9, 11), np.float32)
    mask = np.ones((2, 9, 9, 11), np.float32)
    y = cref.dcn_v2_forward(x, offset, mask, w, b, acc_mode=acc_mode)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(),
                   torch.from_numpy(b).double(), padding=1).float().numpy()
    assert np.abs(y - ref).max() < 2e-5


def test_integer_offsets_are_shifted_taps():
    """Integer (dy,dx) per tap == conv on a shifted, zero-padded input."""
    x = _rand((1, 3, 8, 8), 4)
    w = _rand((2, 3, 3, 3), 5)
    b = np.zeros(2, np.float32)
    rng = np.random.RandomState(6)
    dy, dx = rng.randint(-2, 3, size=9), rng.randint(-2, 3, size=9)
    offset = np.zeros((1, 18, 8, 8), np.float32)
    for k in range(9):
        offset[0, 2 * k] = dy[k]
        offset[0, 2 * k + 1] = dx[k]
    mask = np.ones((1, 9, 8, 8), np.float32)
    y = cref.dcn_v2_forward(x, offset, mask, w, b)
    xp = np.zeros((3, 8 + 8, 8 + 8), np.float64)
    xp[:, 4:12, 4:12] = x[0]
    ref = np.zeros((2, 8, 8))
    for k in range(9):
        i, j = divmod(k, 3)
        for h in range(8):
            for ww in range(8):
                sy, sx = h - 1 + i + dy[k], ww - 1 + j + dx[k]
                ref[:, h, ww] += w[:, :, i, j].astype(np.float64) @ xp[:, sy + 4, sx + 4]
    assert np.abs(y[0] - ref).max() < 1e-5


def test_half_pixel_bilinear_single_tap():
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    w = np.zeros((1, 1, 3, 3), np.float32)
    w[0, 0, 1, 1] = 1.0
    offset = np.zeros((1, 18, 4, 4), np.float32)
    offset[0, 8] = 0.5   # centre tap dy
    offset[0, 9] = 0.25  # centre tap dx
    mask = np.ones((1, 9, 4, 4), np.float32)
    y = cref.dcn_v2_forward(x, offset, mask, w, np.zeros(1, np.float32))
    # interior pixel (1,1): sample at (1.5, 1.25)
    v = 0.5 * (0.75 * x[0, 0, 1, 1] + 0.25 * x[0, 0, 1, 2]) + 0.5 * (0.75 * x[0, 0, 2, 1] + 0.25 * x[0, 0, 2, 2])
    assert abs(y[0, 0, 1, 1] - v) < 1e-6
    # bottom row (3, 1): h_high = 4 is outside -> only the top corners contribute
    v = 0.5 * (0.75 * x[0, 0, 3, 1] + 0.25 * x[0, 0, 3, 2])
    assert abs(y[0, 0, 3, 1] - v) < 1e-6


def test_window_edge_rule():
    """dcn_v2_im2col_cuda.cu:165: sample iff h_im > -1 && w_im > -1 && h_im < H && w_im < W."""
    H = W = 4
    x = np.ones((1, H, W), np.float32)
    offset = np.zeros((18, H, W), np.float32)
    mask = np.ones((9, H, W), np.float32)

    def centre_tap_at(dy, dx):
        off = offset.copy()
        off[8], off[9] = dy, dx
        cols = cref.dcn_v2_im2col(x, off, mask)
        return cols[4]  # centre tap of channel 0

    assert centre_tap_at(-1.0, 0.0)[0, 0] == 0.0         # h_im == -1 exactly: outside
    assert abs(centre_tap_at(-0.75, 0.0)[0, 0] - 0.25) < 1e-7   # h_im = -0.75: only h_high=0 row, weight lh=0.25
    assert centre_tap_at(float(H), 0.0)[0, 0] == 0.0     # h_im == H: outside
    got = centre_tap_at(0.0, 0.5)[0, W - 1]              # w_im = W-0.5 < W: w_high = W dropped
    assert abs(got - 0.5) < 1e-7
    assert centre_tap_at(0.0, 1.0)[0, W - 1] == 0.0      # w_im == W


def test_mask_scales_columns_and_bias_added():
    x = _rand((1, 4, 5, 5), 7)
    w = _rand((3, 4, 3, 3), 8)
    b = _rand((3,), 9)
    offset = _rand((1, 18, 5, 5), 10, 0.7)
    mask = np.random.RandomState(11).uniform(0, 1, (1, 9, 5, 5)).astype(np.float32)
    y1 = cref.dcn_v2_forward(x, offset, mask, w, b)
    y2 = cref.dcn_v2_forward(x, offset, 2 * mask, w, b)
    assert np.allclose(y2 - b[None, :, None, None], 2 * (y1 - b[None, :, None, None]), atol=1e-5)


def test_shape_mismatch_raises():
    with pytest.raises(RuntimeError):
        cref.dcn_v2_forward(np.zeros((1, 4, 4, 4), np.float32), np.zeros((1, 18, 4, 4), np.float32),
                            np.zeros((1, 9, 4, 4), np.float32), np.zeros((2, 3, 3, 3), np.float32),
                            np.zeros(2, np.float32))
