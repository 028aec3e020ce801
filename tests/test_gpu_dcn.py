"""Fused HIP DCNv2 (through the C ABI, NCHW drop-in entry and NHWC native entry) vs
  * the reference's own sampling kernel built test-only (oracle/_ref, when present) and the
    committed fixtures made from it (tests/golden/ref_golden.npz), and
  * the C oracle, itself bit-identical to oracle/_ref (tests/test_oracle_ref.py).
fp32 tolerance: |diff| <= 2e-5 * (1 + |ref|) (fp32 MFMA, different summation order than the
checker's double accumulation)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import cref, ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.gpu
TOL = 2e-5
DCN_FORM_DEFAULT = 0     # cn_set_tuning key 23 as the library starts (csrc/cn_conv.hip g_tune_dcn_form)


def _check(y, ref):
    err = np.abs(y - ref) / (1 + np.abs(ref))
    if not err.max() < TOL:      # say where and how many: a lost tile looks different from a rounding excursion
        idx = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError("max err %.3e at %s (got %r, want %r); %d of %d cells beyond the bar"
                             % (err.max(), idx, float(y[idx]), float(ref[idx]), int((err >= TOL).sum()), err.size))


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
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 20 + Cin)
    want = cref.dcn_v2_forward(x, off, mask, w, b)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)])
    _check(y.cpu().numpy(), want)
    if ref.available():       # the reference's own kernel on the same inputs
        _check(y.cpu().numpy(), ref.dcn_v2_forward(x, off, mask, w, b))


def _gen_ref():
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(GOLDEN, "gen_golden_ref.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dcn_vs_reference_kernel_fixtures(dev):
    """Outputs of the REFERENCE's kernel (oracle/_ref at fixture-generation time) for tuned and
    general-domain configurations: deformable groups, stride 2, dilation 2, 5x5, Cin = 2 / 6 / 7,
    stress offsets."""
    from centernet_amd.dcn_v2 import dcn_v2_forward
    gen = _gen_ref()
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    for name, cfg in gen.DCN_CASES.items():
        x, off, mask, w, b, kw = gen.dcn_inputs(cfg)
        y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)],
                           stride=kw["stride"], padding=kw["pad"], dilation=kw["dil"],
                           deformable_groups=kw["dg"])
        err = np.abs(y.cpu().numpy() - z["dcn_" + name + "_y"]) / (1 + np.abs(z["dcn_" + name + "_y"]))
        assert err.max() < TOL, (name, err.max())


@pytest.mark.parametrize("cfg", [
    dict(B=2, Cin=64, H=24, W=20, Cout=64, dg=2),
    dict(B=2, Cin=12, H=15, W=17, Cout=9, dg=3),
    dict(B=1, Cin=6, H=17, W=13, Cout=70, stride=2),
    dict(B=1, Cin=6, H=17, W=13, Cout=5, stride=2, pad=0),
    dict(B=2, Cin=6, H=19, W=16, Cout=8, dil=2, pad=2),
    dict(B=1, Cin=4, H=14, W=18, Cout=4, k=5, pad=2, dg=2),
    dict(B=1, Cin=5, H=9, W=9, Cout=3, k=1, pad=0),
    dict(B=1, Cin=3, H=33, W=9, Cout=130, k=3),
])
def test_dcn_general_domain_vs_oracle(dev, cfg):
    """Everything outside the tuned configuration (dcn_v2_cuda.c:10-102 takes any kernel /
    stride / pad / dilation / deformable_group): cn_dcn_general.hip through the same entry."""
    from centernet_amd.dcn_v2 import dcn_v2_forward
    B, Cin, H, W, Cout = cfg["B"], cfg["Cin"], cfg["H"], cfg["W"], cfg["Cout"]
    k, stride, pad, dil, dg = cfg.get("k", 3), cfg.get("stride", 1), cfg.get("pad", 1), cfg.get("dil", 1), cfg.get("dg", 1)
    Ho, Wo = cref.out_hw(H, W, k, k, stride, pad, dil)
    x = synth.normal((B, Cin, H, W), 1.0, 3)
    w = synth.normal((Cout, Cin, k, k), (2.0 / (Cin * k * k)) ** 0.5, 4)
    b = synth.normal((Cout,), 0.1, 5)
    off = synth.normal((B, dg * 2 * k * k, Ho, Wo), 2.0, 6)
    off[0, :, 0, :] = -1.0
    mask = synth.uniform((B, dg * k * k, Ho, Wo), 0, 1, 7)
    want = cref.dcn_v2_forward(x, off, mask, w, b, stride, pad, dil, dg)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)], stride=stride,
                       padding=pad, dilation=dil, deformable_groups=dg)
    assert tuple(y.shape) == want.shape
    _check(y.cpu().numpy(), want)
    # mask sigmoid fused (DCN.forward, dcn_v2.py:67)
    logit = synth.normal(mask.shape, 1.0, 8)
    want = cref.dcn_v2_forward(x, off, (1 / (1 + np.exp(-logit.astype(np.float64)))).astype(np.float32),
                               w, b, stride, pad, dil, dg)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, logit, w, b)], stride=stride,
                       padding=pad, dilation=dil, deformable_groups=dg, apply_mask_sigmoid=True)
    _check(y.cpu().numpy(), want)


def test_reference_example_dcn_module_two_deformable_groups(dev):
    """DCNv2/test.py:169-179: DCN(64, 64, 3x3, stride 1, padding 1, deformable_groups=2) on a
    (2, 64, 128, 128) input, un-skipped, vs the oracle chain."""
    from centernet_amd.dcn_v2 import DCN
    from oracle import net_oracle
    m = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2)
    synth.fill_state_dict_(m, 11)
    x = torch.from_numpy(synth.normal((2, 64, 128, 128), 1.0, 12))
    sd = {"d." + k: v for k, v in m.state_dict().items()}
    want = net_oracle.dcn(x, sd, "d", deformable_groups=2).numpy()
    y = m.eval()(x.to(dev)).cpu().numpy()
    assert y.shape == (2, 64, 128, 128)
    _check(y, want)


def test_dcn_module_general_stride_and_odd_channels(dev):
    from centernet_amd.dcn_v2 import DCN
    from oracle import net_oracle
    m = DCN(6, 10, kernel_size=(3, 3), stride=2, padding=1, deformable_groups=3)
    synth.fill_state_dict_(m, 13)
    x = torch.from_numpy(synth.normal((2, 6, 21, 18), 1.0, 14))
    sd = {"d." + k: v for k, v in m.state_dict().items()}
    want = net_oracle.dcn(x, sd, "d", stride=2, deformable_groups=3).numpy()
    y = m.eval()(x.to(dev)).cpu().numpy()
    _check(y, want)


def test_reference_zero_offset_identity_kat_exact_shape(dev):
    """DCNv2/test.py:16-19,32-65 at the reference's own shape N, inC, inH, inW = 2, 2, 4, 4."""
    from centernet_amd.dcn_v2 import DCNv2
    N, C, H, W = 2, 2, 4, 4
    x = synth.normal((N, C, H, W), 1.0, 0)
    m = DCNv2(C, C, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(dev)
    with torch.no_grad():
        m.weight.zero_()
        m.weight[torch.arange(C), torch.arange(C), 1, 1] = 1.0
        m.bias.zero_()
    y = m(torch.from_numpy(x).to(dev), torch.zeros((N, 18, H, W), device=dev),
          torch.full((N, 9, H, W), 0.5, device=dev))
    assert np.abs(2 * y.cpu().numpy() - x).max() < 1e-6


def test_reference_zero_offset_identity_kat(dev):
    """DCNv2/test.py:32-65 on the tuned path (Cin = 4: 16-byte channel vectors)."""
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
    want = ref.dcn_v2_forward(x, off, mask, w, b) if ref.available() else \
        cref.dcn_v2_forward(x, off, mask, w, b)
    y = dcn_v2_forward(*[torch.from_numpy(a).to(dev) for a in (x, off, mask, w, b)])
    _check(y.cpu().numpy(), want)


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
    want = net_oracle.dcn(x, sd, "d").numpy()
    y = m.eval()(x.to(dev)).cpu().numpy()
    _check(y, want)


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


# ---- the kernel the benchmark runs: cn_dcn_v2_forward_nhwc with CN_DTYPE_F32S (NHWC input,
# f32s-packed weight, tap split 1 / 3 / 9), entered with explicit offsets and masks
def _dcn_f32s_nhwc(dev, x, off, mask, w, b, tap_split, out_plain, form=1, msig=False):
    """x (B,C,H,W), off (B,18,H,W), mask (B,9,H,W) numpy -> (B,Cout,H,W) numpy through
    PlanBuilder.dcn(om=...) = cn_dcn_v2_forward_nhwc(dtype = CN_DTYPE_F32S).  msig: hand the mask
    over as logits and let the kernel apply the sigmoid (dcn_v2.py:67: the instantiation the
    networks run)."""
    from centernet_amd import native
    from centernet_amd.dcn_v2 import DCNv2
    from centernet_amd.engine import PlanBuilder, Act, exponent_for
    B, C, H, W = x.shape
    Co = w.shape[0]
    m = DCNv2(C, Co, (3, 3), 1, 1)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w))
        m.bias.copy_(torch.from_numpy(b))
    m.conv_offset_mask = None
    om = np.zeros((B, H, W, 32), np.float32)
    om[..., :18] = off.transpose(0, 2, 3, 1)
    om[..., 18:27] = mask.transpose(0, 2, 3, 1)
    if msig:
        m64 = om[..., 18:27].astype(np.float64)
        om[..., 18:27] = np.log(m64 / (1.0 - m64)).astype(np.float32)
    want_scale = float(np.abs(x).max())
    lib = native.lib()
    lib.cn_set_tuning(13, tap_split)
    lib.cn_set_tuning(23, form)    # 1 = global-gather form, 2 = register-sampling window form, 4 / 5 = team form (T / N mode), 0 = by shape
    try:
        pb = PlanBuilder(dev, B, H, W, split=True,
                         exps={"x": exponent_for(want_scale), "t1": exponent_for(4.0 * want_scale)})
        xa = Act(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev), B, H, W, C,
                 exp=pb._exp("x"), lid="x")
        oma = Act(torch.from_numpy(om).to(dev), B, H, W, 27, pitch=32)
        y = pb.dcn(xa, m, om=oma, mask_sigmoid=bool(msig), out_plain=out_plain)
        assert y.fmt == ("f32" if out_plain else "f32s")
        for op in pb.ops:
            op()
        torch.cuda.synchronize()
    finally:
        lib.cn_set_tuning(13, 0)
        lib.cn_set_tuning(23, DCN_FORM_DEFAULT)
    return y.to_float().permute(0, 3, 1, 2).cpu().numpy()


@pytest.mark.parametrize("tap_split", [1, 3, 9])
def test_f32s_nhwc_kernel_vs_reference_kernel_fixtures(dev, tap_split):
    """The product kernel against outputs of the REFERENCE's own kernel (ref_golden.npz) for
    every fixture in its domain (3x3 / stride 1 / pad 1 / one group, Cin % 4 == 0), stress
    offsets included."""
    gen = _gen_ref()
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    ran = 0
    for name, cfg in gen.DCN_CASES.items():
        if (cfg["k"], cfg["stride"], cfg["pad"], cfg["dil"], cfg["dg"]) != (3, 1, 1, 1, 1) or cfg["Cin"] % 4:
            continue
        x, off, mask, w, b, _ = gen.dcn_inputs(cfg)
        for out_plain in (False, True):
            y = _dcn_f32s_nhwc(dev, x, off, mask, w, b, tap_split, out_plain)
            want = z["dcn_" + name + "_y"]
            err = np.abs(y - want) / (1 + np.abs(want))
            assert err.max() < TOL, (name, tap_split, out_plain, err.max())
        ran += 1
    assert ran >= 3


def test_f32s_nhwc_kernel_stress_offsets(dev):
    """Offsets ~ U(-H, H) and exactly -1 / H / integers (dcn_v2_im2col_cuda.cu:165, :30-41)."""
    B, Cin, H, W, Cout = 2, 64, 12, 12, 64
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 77)
    off = synth.uniform((B, 18, H, W), -H, H, 78)
    off[0, :, 0, :] = -1.0
    off[0, :, 1, :] = float(H)
    off[1, :, 2, :] = np.round(off[1, :, 2, :])
    want = ref.dcn_v2_forward(x, off, mask, w, b) if ref.available() else \
        cref.dcn_v2_forward(x, off, mask, w, b)
    for tap_split in (1, 3, 9):
        _check(_dcn_f32s_nhwc(dev, x, off, mask, w, b, tap_split, False), want)


@pytest.mark.parametrize("form", [0, 1, 4, 5, 6, 7])
@pytest.mark.parametrize("shape", [(512, 16, 256), (256, 32, 128), (128, 64, 64),     # resdcn_18
                                   (64, 128, 64), (128, 64, 128), (256, 32, 256),      # dla_34
                                   (256, 32, 64)])
def test_f32s_nhwc_kernel_at_benchmark_batch(dev, shape, form):
    """B = 32, the layer shapes of resdcn_18 and dla_34: the launch the benchmark
    times -- form 0 = the library's DEFAULT choice for the shape (round 5: the team form of the
    window kernel, csrc/cn_dcn3.hip, N mode where Cout is a multiple of 128 and the grid still fills
    the chip, K-split on the 16^2 map), form 1 = the global-gather kernel, 4 / 5 = the team form
    forced into T / N mode -- default tap split; images 0, 13 and 31 against the C oracle (the
    operator is per image, dcn_v2_cuda.c:61)."""
    Cin, HW, Cout = shape
    B = 32
    if form >= 6 and Cout % 128:
        pytest.skip("the wide form takes Cout % 128 == 0")
    x, off, mask, w, b = _case(B, Cin, HW, HW, Cout, 300 + Cin)
    y = _dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, False, form=form)
    for i in (0, 13, 31):
        want = cref.dcn_v2_forward(x[i:i + 1], off[i:i + 1], mask[i:i + 1], w, b)
        _check(y[i:i + 1], want)


# ---- the LDS-window forms (input window in LDS, every lane samples its own MFMA operand, 8 x 16 pixel
# tiles): form 2 = csrc/cn_dcn2.hip dcn_reg_kernel (four waves per tile), forms 4 / 5 = csrc/cn_dcn3.hip
# dcn_team_kernel (eight waves per tile in two teams: T mode = the teams split the (tap, chunk) steps
# of a 64-channel block and add up, N mode = each takes 64 of 128 output channels; window by LDS-DMA)
# forms 6 / 7 = csrc/cn_dcn4.hip dcn_wide_kernel (round 6: a workgroup owns ALL output channels of its tile,
# 8 or 4 blocks of 32 -- every sample formed once; weights through an LDS ring; Cout % 128 == 0)
WINDOW_FORMS = [2, 4, 5, 6, 7]


def _window_takes(form, Cin, H, W, Cout):
    if form >= 6 and Cout % 128:
        return False
    return Cin % 32 == 0 and H % 8 == 0 and W % 16 == 0 and Cout > 32 and Cout % 4 == 0


@pytest.mark.parametrize("form", WINDOW_FORMS)
def test_f32s_window_kernel_vs_reference_kernel_fixtures(dev, form):
    gen = _gen_ref()
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    ran = 0
    for name, cfg in gen.DCN_CASES.items():
        if (cfg["k"], cfg["stride"], cfg["pad"], cfg["dil"], cfg["dg"]) != (3, 1, 1, 1, 1):
            continue
        if not _window_takes(form, cfg["Cin"], cfg["H"], cfg["W"], cfg["Cout"]):
            continue
        x, off, mask, w, b, _ = gen.dcn_inputs(cfg)
        for out_plain in (False, True):
            y = _dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, out_plain, form=form)
            want = z["dcn_" + name + "_y"]
            err = np.abs(y - want) / (1 + np.abs(want))
            assert err.max() < TOL, (name, out_plain, err.max())
        ran += 1
    assert ran >= (1 if form < 6 else 0)     # no reference fixture has Cout % 128 == 0 (the wide form's domain)


@pytest.mark.parametrize("form", WINDOW_FORMS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16, 64), (1, 128, 24, 48, 128), (3, 96, 8, 16, 64),
                                   (1, 512, 16, 16, 256), (2, 32, 16, 32, 36), (1, 64, 32, 32, 100),
                                   (1, 64, 8, 24, 192), (2, 64, 16, 32, 256), (1, 96, 8, 16, 384)])
@pytest.mark.parametrize("off_std", [0.5, 2.0, 6.0])
def test_f32s_window_kernel_vs_oracle(dev, shape, off_std, form):
    """Offsets well inside the window (0.5 px), around its reach (2 px: a few per cent of the samples
    take the global fallback) and mostly beyond it (6 px); several chunks, two N tiles, ragged
    Cout, non-square maps, map = one tile."""
    B, Cin, H, W, Cout = shape
    if not _window_takes(form, Cin, H, W, Cout):
        pytest.skip("shape outside this form's domain")
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 500 + Cin + H, off_std=off_std)
    want = cref.dcn_v2_forward(x, off, mask, w, b)
    for out_plain in (False, True):
        _check(_dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, out_plain, form=form), want)


@pytest.mark.parametrize("form", WINDOW_FORMS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16, 64), (1, 128, 24, 48, 128), (1, 512, 16, 16, 256),
                                   (2, 96, 32, 32, 68)])
@pytest.mark.parametrize("off_std", [1.0, 5.0])
def test_f32s_window_kernel_with_the_sigmoid_inside(dev, shape, off_std, form):
    """The instantiation the networks run: mask logits in, sigmoid in the kernel (dcn_v2.py:67); the
    team form folds sigmoid(mask) * 2^-e into the four corner weights and neither clamps nor tracks
    per sample."""
    B, Cin, H, W, Cout = shape
    if not _window_takes(form, Cin, H, W, Cout):
        pytest.skip("shape outside this form's domain")
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 700 + Cin + H, off_std=off_std)
    want = cref.dcn_v2_forward(x, off, mask, w, b)
    for out_plain in (False, True):
        _check(_dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, out_plain, form=form, msig=True), want)


@pytest.mark.parametrize("form", WINDOW_FORMS)
def test_f32s_window_kernel_stress_offsets(dev, form):
    """Offsets ~ U(-H, H) and exactly -1 / H / integers (dcn_v2_im2col_cuda.cu:165, :30-41): every
    sample takes the fallback or lies on a rule boundary."""
    B, Cin, H, W, Cout = 2, 64, 16, 16, (64 if form < 6 else 128)
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 77)
    off = synth.uniform((B, 18, H, W), -H, H, 78)
    off[0, :, 0, :] = -1.0
    off[0, :, 1, :] = float(H)
    off[1, :, 2, :] = np.round(off[1, :, 2, :])
    off[1, :, 3, :] = 0.0
    off[1, 0::2, 4, :] = -3.0          # exactly on the window's reach
    off[1, 1::2, 4, :] = 2.999
    want = ref.dcn_v2_forward(x, off, mask, w, b) if ref.available() else \
        cref.dcn_v2_forward(x, off, mask, w, b)
    _check(_dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, False, form=form), want)
    _check(_dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, False, form=form, msig=True), want)


@pytest.mark.parametrize("form", WINDOW_FORMS)
def test_f32s_window_kernel_is_run_to_run_deterministic_at_benchmark_batch(dev, form):
    """B = 32 (a launch of the size the benchmark times): identical bits over repeated launches."""
    B, Cin, HW, Cout = (32, 128, 64, 64) if form < 6 else (32, 256, 32, 256)
    x, off, mask, w, b = _case(B, Cin, HW, HW, Cout, 900)
    first = None
    for _ in range(4):
        y = _dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, False, form=form)
        if first is None:
            first = y
            for i in (0, 17, 31):
                _check(y[i:i + 1], cref.dcn_v2_forward(x[i:i + 1], off[i:i + 1], mask[i:i + 1], w, b))
        assert np.array_equal(y, first)


def test_f32s_window_kernel_k_split_small_map(dev):
    """A map with too few tiles for the chip but a deep K (the 512 -> 256 @ 16 x 16 layer of resdcn_18):
    under the default form selection the register-sampling kernel splits the 32-channel chunks
    over 8 workgroups per tile (raw partial slabs + fixed-order reduce): against the oracle, both
    output formats, and bit-identical over repeated launches."""
    B, Cin, H, W, Cout = 8, 512, 16, 16, 256
    x, off, mask, w, b = _case(B, Cin, H, W, Cout, 4242, off_std=1.5)
    want = cref.dcn_v2_forward(x, off, mask, w, b)
    first = None
    for out_plain in (False, True, False):
        y = _dcn_f32s_nhwc(dev, x, off, mask, w, b, 0, out_plain, form=0)
        _check(y, want)
        if not out_plain:
            if first is None:
                first = y
            assert np.array_equal(y, first)
