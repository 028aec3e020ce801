"""Fused HIP DCNv2 (through the C ABI, NCHW drop-in entry and NHWC native entry) vs the C
oracle.  fp32 tolerance: |diff| <= 2e-5 * (1 + |ref|) (fp32 MFMA, different summation
order than the oracle's double accumulation)."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import cref

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _check(y, ref):
    err = np.abs(y - ref) / (1 + np.abs(ref))
    assert err.max() < TOL, err.max()


def _case(B, Cin, H, W, Cout, seed, off_std=2.0):
    x = synth.normal((B, Cin, H, W), 1.0, seed)
    w = synth.normal((Cout, Cin, 3, 3), (2.0 / (Cin * 9)) ** 0.5, seed + 1)
    b = synth.normal((Cout,), 0.1, seed + 2)
    off = synth.normal((B, 18, H, W), off_std, seed + 3)
    mask = 1.0 / (1.0 + np.exp(-synth.normal((B, 9, H, W), 1.0, seed + 4)))
    return x, off, mask.astype(np.float32), w, b


@pytest.mark.parametrize("shape", [(2, 2, 4, 4, 2), (2, 64, 16, 16, 64), (1, 512, 16, 16, 256),
                                   (2, 256, 32, 32, 128), (1, 128, 64, 64, 64), (1, 8, 5, 7, 12),
                                   (3, 36, 9, 11, 40)])
def test_dcn_nchw_entry_vs_oracle(dev, shape):
    from centernet_amd.dcn_v2 import dcn_v2_forward
    B, Cin, H, W, Cout = shape
    if Cin % 4:
        pytest.skip("HIP path needs Cin % 4 == 0")
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 20 + Cin)
    ref = cref.dcn_v2_forward(x, off, mask, w, b)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)])
    _check(y.cpu().numpy(), ref)


def test_reference_zero_offset_identity_kat(dev):
    """DCNv2/test.py:32-65 (shapes widened to Cin=4 for the 16-byte channel vectors)."""
    from centernet_amd.dcn_v2 import dcn_v2_forward
    N, C, H, W = 2, 4, 4, 4
    x = synth.normal((N, C, H, W), 1.0, 0)
    w = np.zeros((C, C, 3, 3), np.float32)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    y = dcn_v2_forward(torch.from_numpy(x).to(dev), torch.zeros((N, 18, H, W), device=dev),
                       torch.full((N, 9, H, W), 0.5, device=dev), torch.from_numpy(w).to(dev),
                       torch.zeros(C, device=dev))
    assert np.abs(2 * y.cpu().numpy() - x).max() < 1e-6


def test_stress_offsets_far_outside(dev):
    """Offsets ~ U(-H, H): most samples fall on or outside the border rule."""
    from centernet_amd.dcn_v2 import dcn_v2_forward
    B, Cin, H, W, Cout = 2, 32, 12, 12, 32
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 77)
    off = synth.uniform((B, 18, H, W), -H, H, 78)
    # and exact integers / exact -1 / exact H to hit the comparisons
    off[0, :, 0, :] = -1.0
    off[0, :, 1, :] = float(H)
    off[1, :, 2, :] = np.round(off[1, :, 2, :])
    ref = cref.dcn_v2_forward(x, off, mask, w, b)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)])
    _check(y.cpu().numpy(), ref)


def test_dcn_module_with_offset_conv(dev):
    """DCN module: conv_offset_mask + sigmoid + deformable conv, vs oracle chain."""
    from centernet_amd.dcn_v2 import DCN
    from oracle import net_oracle
    torch.manual_seed(0)
    m = DCN(64, 32, (3, 3), 1, 1)
    synth.fill_state_dict_(m, 5)
    x = synth.images(2, 16, 16, 3)[:, :1].repeat(1, 64, 1, 1) * synth.normal((1, 64, 1, 1), 1.0, 9)
    x = x.contiguous()
    sd = {"d." + k: v for k, v in m.state_dict().items()}
    ref = net_oracle.dcn(x, sd, "d").numpy()
    y = m.eval()(x.to(dev)).cpu().numpy()
    _check(y, ref)


def test_full_size_linearity(dev):
    """BASELINE size (B=32, resdcn_18 layer 128->64 @ 64x64): the op is linear in the
    input and in the mask for fixed offsets."""
    from centernet_amd.dcn_v2 import dcn_v2_forward
    B, Cin, H, W, Cout = 32, 128, 64, 64, 64
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn((B, Cin, H, W), generator=g).to(dev)
    x2 = torch.randn((B, Cin, H, W), generator=g).to(dev)
    off = (2 * torch.randn((B, 18, H, W), generator=g)).to(dev)
    mask = torch.rand((B, 9, H, W), generator=g).to(dev)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.03).to(dev)
    b0 = torch.zeros(Cout, device=dev)
    y1 = dcn_v2_forward(x1, off, mask, w, b0)
    y2 = dcn_v2_forward(x2, off, mask, w, b0)
    y12 = dcn_v2_forward(x1 + x2, off, mask, w, b0)
    assert float((y12 - (y1 + y2)).abs().max()) < 1e-4
    ym = dcn_v2_forward(x1, off, 0.5 * mask, w, b0)
    assert float((ym - 0.5 * y1).abs().max()) < 1e-5


def test_lds_window_variant_matches_oracle(dev):
    """The opt-in LDS-window kernel (cn_dcn.hip, cn_set_tuning key 11) against the oracle,
    including offsets far beyond its window (global fallback per corner)."""
    from centernet_amd import native
    from centernet_amd.dcn_v2 import dcn_v2_forward
    lib = native.lib()
    try:
        assert lib.cn_set_tuning(11, 1) == 0
        for (B, Cin, H, W, Cout, std) in [(2, 64, 16, 16, 64, 2.0), (1, 128, 20, 12, 128, 6.0),
                                          (1, 36, 9, 11, 40, 1.0)]:
            x, off, mask, w, b = _case(B, Cin, H, W, Cout, 50 + Cin, off_std=std)
            ref = cref.dcn_v2_forward(x, off, mask, w, b)
            y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)])
            _check(y.cpu().numpy(), ref)
    finally:
        lib.cn_set_tuning(11, 0)
