"""Implicit-GEMM conv / conv-transpose / max-pool kernels (through the plan builder, i.e.
the C ABI) vs torch CPU fp32 (the reference's third-party arithmetic).  Tolerance:
|diff| <= 2e-5 * (1 + |ref|)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from centernet_amd import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _check(y, ref):
    err = (y - ref).abs() / (1 + ref.abs())
    assert float(err.max()) < TOL, float(err.max())


def _nhwc_act(x, dev):
    from centernet_amd.engine import Act
    B, C, H, W = x.shape
    return Act(x.permute(0, 2, 3, 1).contiguous().to(dev), B, H, W, C)


def _run(pb):
    for op in pb.ops:
        op()
    torch.cuda.synchronize()


def _bn(C, seed):
    bn = torch.nn.BatchNorm2d(C)
    synth.fill_state_dict_(bn, seed)
    return bn.eval()


@pytest.mark.parametrize("cfg", [
    # B, Cin, H, W, Cout, k, stride, pad, bias, bn, relu, residual
    (2, 64, 32, 32, 64, 3, 1, 1, False, True, True, True),
    (2, 64, 32, 32, 128, 3, 2, 1, False, True, True, False),
    (1, 64, 17, 19, 128, 1, 2, 0, False, True, False, False),
    (2, 256, 16, 16, 512, 3, 1, 1, False, True, True, True),
    (1, 512, 16, 16, 27, 3, 1, 1, True, False, False, False),
    (3, 36, 9, 11, 40, 3, 1, 1, True, False, True, False),
    (1, 64, 128, 128, 192, 3, 1, 1, True, False, True, False),
    (2, 128, 8, 8, 256, 1, 1, 0, False, True, True, True),
    # small maps with deep K: the split-K path (several workgroups per output tile + reduce)
    (1, 512, 16, 16, 512, 3, 1, 1, False, True, True, True),
    (2, 512, 8, 8, 27, 3, 1, 1, True, False, False, False),
    (1, 384, 4, 4, 384, 3, 1, 1, False, True, True, False),
    (2, 256, 16, 16, 128, 3, 2, 1, False, True, True, False),
    (5, 48, 100, 132, 64, 3, 1, 1, True, False, True, False),
    # partially filled last K chunk (DLA level0: 16 -> 16): empty 8-channel groups are skipped
    (2, 16, 64, 64, 16, 3, 1, 1, False, True, True, False),
    (1, 40, 20, 24, 24, 3, 1, 1, True, False, False, True),
    # 16-channel input on 128-pixel-multiple rows: the 16-channel kernels (cn_conv16.hip: fp32
    # 16x16x4 MFMA / f32s 16x16x32 f16 MFMA with plain tensors on both sides)
    (2, 16, 20, 128, 16, 3, 1, 1, False, True, True, False),    # DLA level0 form
    (1, 16, 33, 256, 32, 3, 2, 1, False, True, True, False),    # DLA level1 form (stride 2)
    (2, 16, 12, 256, 24, 3, 1, 1, True, False, False, False),   # two N blocks, ragged Cout
    (3, 16, 9, 256, 9, 3, 2, 1, True, False, True, False),      # one N block, stride 2
    (5, 16, 130, 128, 16, 3, 1, 1, False, True, True, False),   # 650 tiles > 512 workgroups
])
@pytest.mark.parametrize("split", [True, False], ids=["f32s", "fp32mfma"])
def test_conv_bn_relu_residual(dev, cfg, split):
    _conv_case(dev, cfg, split)


def test_conv_256_pixel_tiles(dev):
    """Opt-in 8 x 32 tiles of the LDS-halo kernel (cn_set_tuning key 14)."""
    from centernet_amd import native
    lib = native.lib()
    lib.cn_set_tuning(14, 1)
    try:
        _conv_case(dev, (16, 64, 128, 128, 64, 3, 1, 1, False, True, True, True))
        _conv_case(dev, (17, 48, 100, 132, 64, 3, 1, 1, True, False, True, False))
    finally:
        lib.cn_set_tuning(14, 0)


def test_conv_four_workgroups_per_cu_variant(dev):
    """64-wide halo tiles at four workgroups per CU (single-buffered weights; key 19)."""
    from centernet_amd import native
    lib = native.lib()
    try:
        for v in (1, 2):
            lib.cn_set_tuning(19, v)
            _conv_case(dev, (4, 64, 64, 64, 64, 3, 1, 1, False, True, True, True))
            _conv_case(dev, (3, 96, 19, 27, 40, 3, 1, 1, True, False, True, False))
    finally:
        lib.cn_set_tuning(19, 0)


def test_conv_8_wave_tiles(dev):
    """Both workgroup shapes of the 128-wide LDS-halo tiles (key 15: 8 waves default, 4 waves)."""
    from centernet_amd import native
    lib = native.lib()
    try:
        for v in (1, 0):
            lib.cn_set_tuning(15, v)
            _conv_case(dev, (2, 256, 32, 32, 256, 3, 1, 1, False, True, True, True))
            _conv_case(dev, (3, 160, 19, 27, 130, 3, 1, 1, True, False, True, False))
    finally:
        lib.cn_set_tuning(15, 1)


def _conv_case(dev, cfg, split=False):
    """``split``: f32s (three fp16 MFMAs per product) or the plain fp32 matrix instruction;
    the knob tests below exercise the fp32-MFMA tile variants."""
    from centernet_amd.engine import PlanBuilder
    B, Cin, H, W, Cout, k, s, p, use_bias, use_bn, relu, use_res = cfg
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1))
    w = torch.from_numpy(synth.normal((Cout, Cin, k, k), (2.0 / (Cin * k * k)) ** 0.5, 2))
    bias = torch.from_numpy(synth.normal((Cout,), 0.3, 3)) if use_bias else None
    bn = _bn(Cout, 4) if use_bn else None
    ref = F.conv2d(x, w, bias, s, p)
    if bn is not None:
        ref = bn(ref)
    res = None
    if use_res:
        res = torch.from_numpy(synth.normal(tuple(ref.shape), 1.0, 5))
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    pb = PlanBuilder(dev, B, H, W, split=split)
    y = pb.conv(_nhwc_act(x, dev), w, bias=bias, bn=bn, relu=relu,
                residual=_nhwc_act(res, dev) if use_res else None, stride=s, padding=p)
    # (the 16-channel layers compute in f32s but keep plain tensors on both sides)
    want_s = split and PlanBuilder._f32s_conv_form(k, k, s, p, 1, False, Cin, Cout) and \
        not PlanBuilder._narrow_form(k, k, p, 1, False, Cin, Cout)
    assert y.fmt == ("f32s" if want_s else "f32")
    y = pb.plain(y)
    _run(pb)
    _check(y.t[..., :Cout].permute(0, 3, 1, 2).cpu(), ref.detach())
    if split:
        # f32s activations in, f32s residual, plain out: the other format combinations
        pb = PlanBuilder(dev, B, H, W, split=True)
        xs = pb.packed(_nhwc_act(x, dev))
        rs = pb.packed(_nhwc_act(res, dev)) if use_res else None
        y = pb.conv(xs, w, bias=bias, bn=bn, relu=relu, residual=rs, stride=s, padding=p,
                    out_plain=True)
        assert y.fmt == "f32"
        _run(pb)
        _check(y.t[..., :Cout].permute(0, 3, 1, 2).cpu(), ref.detach())


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cout, one launch?
    (48, 64, 256, 64, True),      # one tile per row, strips of 4 pooled rows
    (12, 256, 512, 64, True),     # two tiles per row: the column that crosses the tile edge
    (24, 512, 256, 48, True),     # strips of 8 pooled rows, ragged Cout
    (24, 128, 768, 64, True),     # three tiles per row
    (24, 128, 1024, 64, True),    # four tiles per row
    (2, 64, 256, 64, False),      # too few strips to fill the chip: stem + pooling kernel
    (4, 64, 192, 64, False),      # rows that are not whole 128-pixel tiles
])
def test_stem_with_fused_maxpool_equals_two_launches(dev, cfg, monkeypatch):
    """conv1 -> bn1 -> relu -> maxpool (resnet_dcn.py:138-141,247-250) in ONE kernel: max() is
    exact, so the fused result is bit-identical to the f32s stem followed by the pooling kernel;
    shapes the fused kernel does not take fall back to the two launches."""
    from centernet_amd.engine import PlanBuilder
    B, H, W, Cout, expect_fused = cfg
    x = synth.images(B, H, W, 3)
    w = torch.from_numpy(synth.normal((Cout, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
    bn = _bn(Cout, 4)

    def build(fuse, y_f32s="0"):
        monkeypatch.setenv("CN_FUSE_STEM_POOL", "1" if fuse else "0")
        monkeypatch.setenv("CN_STEM_Y_F32S", y_f32s)
        pb = PlanBuilder(dev, B, H, W, split=True)
        y = pb.conv(pb.set_input(3), w, bn=bn, relu=True, stride=2, padding=3, pool=(3, 2, 1))
        pb.input.t = x.to(dev)
        _run(pb)
        return y, len(pb.ops)
    (two, n2), (one, n1) = build(False), build(True)
    assert n2 == 2
    assert n1 == (1 if expect_fused else 2), (n1, cfg)
    assert one.fmt == two.fmt == "f32" and one.t.shape == two.t.shape == (B, H // 4, W // 4, Cout)
    assert torch.equal(one.t, two.t)
    nb = min(B, 2)
    ref = F.max_pool2d(F.relu(bn(F.conv2d(x[:nb], w, None, 2, 3))), 3, 2, 1).detach()
    _check(one.t[:nb].permute(0, 3, 1, 2).cpu(), ref)
    # the default form: the pooled map written as an f32s tensor (the consumers are f32s layers) --
    # the same values, each as its fp16 (high, low) pair, pad channels of the last group zero
    if expect_fused and Cout % 32 == 0:
        ys, ns = build(True, "1")
        assert ns == 1 and ys.fmt == "f32s"
        h = one.t.half()
        want = h.float() + (one.t - h.float()).half().float()
        assert torch.equal(ys.to_float(), want)


def test_stem_conv_nchw_input(dev):
    from centernet_amd.engine import PlanBuilder
    B, H, W = 2, 64, 96
    x = synth.images(B, H, W, 3)
    w = torch.from_numpy(synth.normal((64, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
    bn = _bn(64, 4)
    ref = F.relu(bn(F.conv2d(x, w, None, 2, 3))).detach()
    pb = PlanBuilder(dev, B, H, W)
    xin = pb.set_input(3)
    y = pb.conv(xin, w, bn=bn, relu=True, stride=2, padding=3)
    pb.input.t = x.to(dev)
    _run(pb)
    _check(y.t.permute(0, 3, 1, 2).cpu(), ref)
    pooled = F.max_pool2d(ref, 3, 2, 1)
    pb2 = PlanBuilder(dev, B, H, W)
    z = pb2.maxpool(_nhwc_act(ref, dev), 3, 2, 1)
    _run(pb2)
    assert torch.equal(z.t.permute(0, 3, 1, 2).cpu(), pooled)


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cout, stride, persistent-eligible (output rows a multiple of 128 px)
    (2, 20, 256, 64, 2, True),     # resnet / hourglass stem shape class (Wo = 128)
    (3, 9, 512, 128, 2, True),     # Wo = 256: two tiles per row, two N tiles (hourglass: 128 ch)
    (2, 11, 128, 16, 1, True),     # DLA base_layer: stride 1, 16 channels (f32s: stem16s_kernel)
    (1, 5, 256, 9, 1, True),       # the same form: two tiles per row, ragged channel count
    (4, 140, 128, 16, 1, True),    # 560 tiles > 512 persistent workgroups
    (1, 40, 256, 12, 2, True),     # <= 16 channels, stride 2, ragged channel count
    (2, 9, 128, 24, 1, True),      # 17..32 channels: 32-wide N tile
    (4, 300, 256, 64, 2, True),    # 600 tiles > 512 persistent workgroups: the tile loop runs > 1x
    (2, 33, 200, 64, 2, False),    # not a multiple of 128: one-tile-per-workgroup kernel
])
def test_stem_kernels_vs_torch(dev, cfg):
    """7x7 stem on the NCHW image: persistent prefetching kernel and the per-tile kernel
    (cn_set_tuning key 12) against torch CPU conv + BN + ReLU."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    B, H, W, Cout, stride, _ = cfg
    x = synth.images(B, H, W, 7)
    w = torch.from_numpy(synth.normal((Cout, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
    bn = _bn(Cout, 4)
    ref = F.relu(bn(F.conv2d(x, w, None, stride, 3))).detach()
    lib = native.lib()
    try:
        for variant in (1, 0):
            lib.cn_set_tuning(12, variant)
            pb = PlanBuilder(dev, B, H, W)
            xin = pb.set_input(3)
            y = pb.conv(xin, w, bn=bn, relu=True, stride=stride, padding=3)
            pb.input.t = x.to(dev)
            _run(pb)
            _check(y.t.permute(0, 3, 1, 2).cpu(), ref)
    finally:
        lib.cn_set_tuning(12, 1)


def test_stem16_f32s_and_fp32_forms(dev):
    """DLA base_layer (7x7 / stride 1, 3 -> 16): the f32s kernel (two shifted copies of the fp16
    window, v_mfma_f32_16x16x32_f16; cn_set_tuning key 27 = 1, default) and the fp32 16x16x4 kernel
    (key 27 = 0) against torch, on an image whose borders matter (7 rows only) and at a scale far
    from 1 (the image exponent is applied while the window is split)."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder, exponent_for
    lib = native.lib()
    try:
        for form in (1, 0):
            assert lib.cn_set_tuning(27, form) == 0
            for (B, H, W, Cout, mul) in [(2, 7, 128, 16, 1.0), (1, 33, 384, 13, 300.0)]:
                x = synth.images(B, H, W, 7) * mul
                w = torch.from_numpy(synth.normal((Cout, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
                bn = _bn(Cout, 4)
                ref = F.relu(bn(F.conv2d(x, w, None, 1, 3))).detach()
                pb = PlanBuilder(dev, B, H, W, split=True, exps={"input": exponent_for(float(x.abs().max()))})
                y = pb.conv(pb.set_input(3), w, bn=bn, relu=True, stride=1, padding=3)
                pb.input.t = x.to(dev)
                _run(pb)
                got = y.t.permute(0, 3, 1, 2).cpu()
                err = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
                assert err < 2e-5, (form, B, H, W, Cout, err)
    finally:
        lib.cn_set_tuning(27, 1)


@pytest.mark.parametrize("split", [True, False], ids=["f32s", "fp32mfma"])
@pytest.mark.parametrize("halo", [True, False])
@pytest.mark.parametrize("cfg", [(2, 64, 16, 16, 64), (1, 256, 16, 16, 256), (1, 128, 9, 7, 64),
                                 (2, 128, 37, 45, 96), (1, 64, 8, 40, 24)])
def test_conv_transpose_4x4_s2(dev, cfg, halo, split):
    """Both forms: LDS-halo parity kernel (default) and the generic implicit GEMM (key 10)."""
    from centernet_amd import native
    native.lib().cn_set_tuning(10, 0 if halo else 1)
    try:
        _deconv_case(dev, cfg, split)
    finally:
        native.lib().cn_set_tuning(10, 0)


def _deconv_case(dev, cfg, split=False):
    from centernet_amd.engine import PlanBuilder
    B, Cin, H, W, Cout = cfg
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1))
    w = torch.from_numpy(synth.normal((Cin, Cout, 4, 4), (2.0 / (Cin * 4)) ** 0.5, 2))
    bn = _bn(Cout, 3)
    ref = F.relu(bn(F.conv_transpose2d(x, w, None, 2, 1, 0))).detach()
    pb = PlanBuilder(dev, B, H, W, split=split)
    y = pb.plain(pb.conv_transpose4x4s2(_nhwc_act(x, dev), w, bn=bn, relu=True))
    _run(pb)
    _check(y.t[..., :Cout].permute(0, 3, 1, 2).cpu(), ref)
    if split:    # f32s input, plain output (the producer of a deformable layer's input)
        pb = PlanBuilder(dev, B, H, W, split=True)
        y = pb.conv_transpose4x4s2(pb.packed(_nhwc_act(x, dev)), w, bn=bn, relu=True, out_plain=True)
        assert y.fmt == "f32"
        _run(pb)
        _check(y.t[..., :Cout].permute(0, 3, 1, 2).cpu(), ref)


@pytest.mark.parametrize("cfg", [(2, 64, 16, 16, 64), (1, 256, 16, 16, 256), (1, 128, 9, 7, 64),
                                 (2, 128, 37, 45, 96), (1, 96, 8, 40, 32), (32, 128, 64, 64, 64)])
def test_conv_transpose_4x4_s2_on_the_persistent_kernel(dev, cfg):
    """ConvTranspose2d(4, 2, 1) in parity form on the persistent loader / consumer kernel
    (cn_conv3x3p.hip, NTAP = 4: item = (tile, parity, output block), four taps per stage): f32s input,
    f32s and plain output, edge tiles, a half-empty output block (96 channels), Cin = 96 -- against
    torch.conv_transpose2d on the CPU and the one-tile-per-workgroup kernel (key 32 = 0); run-to-run
    bit equality.  key 28 = 2 forces the kernel onto shapes with few items; the last case is
    resdcn_18's second up-sampling layer at the benchmark batch, on the library's own routing."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    lib = native.lib()
    B, Cin, H, W, Cout = cfg
    nref = min(B, 2)
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1)).relu_()
    w = torch.from_numpy(synth.normal((Cin, Cout, 4, 4), (2.0 / (Cin * 4)) ** 0.5, 2))
    bn = _bn(Cout, 3)
    ref = F.relu(bn(F.conv_transpose2d(x[:nref], w, None, 2, 1, 0))).detach()
    assert lib.cn_set_tuning(28, 2 if B < 32 else 1) == 0
    try:
        for out_plain in (False, True):
            got = {}
            for key32 in (1, 0):
                assert lib.cn_set_tuning(32, key32) == 0
                pb = PlanBuilder(dev, B, H, W, split=True)
                y = pb.conv_transpose4x4s2(pb.packed(_nhwc_act(x, dev)), w, bn=bn, relu=True, out_plain=out_plain)
                assert y.fmt == ("f32" if out_plain else "f32s")
                _run(pb)
                got[key32] = y.to_float().clone()
                if key32 == 1:
                    raw = y.t.clone()
                    _run(pb)
                    assert torch.equal(raw, y.t), "not deterministic run to run"
                    if not out_plain and Cout % 32 == 0 and y.pitch > Cout:
                        assert float(y.t[..., Cout:].abs().max()) == 0.0
            _check(got[1][:nref].permute(0, 3, 1, 2).cpu(), ref)
            d = float((got[1] - got[0]).abs().max()) / max(1.0, float(got[0].abs().max()))
            assert d < 2e-5, d
    finally:
        lib.cn_set_tuning(28, 1)
        lib.cn_set_tuning(32, 1)


@pytest.mark.parametrize("cfg", [(2, 64, 32, 32, 128), (1, 128, 16, 32, 256), (1, 96, 17, 23, 96),
                                 (2, 64, 40, 24, 64), (3, 32, 9, 70, 32), (32, 64, 128, 128, 128)])
def test_conv3x3_stride2_on_the_persistent_kernel(dev, cfg):
    """3x3 / stride 2 / pad 1 (the first convolution of a down-sampling BasicBlock, resnet_dcn.py:38-67)
    in parity-plane form on the persistent loader / consumer kernel (cn_conv3x3p.hip, S2: four stages of
    4 / 2 / 2 / 1 taps per chunk over the input's parity planes): even and odd map sizes, edge tiles,
    Cin = 96, a half-empty output block, f32s and plain output -- against torch.conv2d on the CPU and
    the implicit-GEMM kernel (key 33 = 0); run-to-run bit equality.  The last case is resdcn_18's
    layer2 entry at the benchmark batch on the library's own routing."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    lib = native.lib()
    B, Cin, H, W, Cout = cfg
    nref = min(B, 2)
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1)).relu_()
    w = torch.from_numpy(synth.normal((Cout, Cin, 3, 3), (2.0 / (Cin * 9)) ** 0.5, 2))
    bn = _bn(Cout, 3)
    ref = F.relu(bn(F.conv2d(x[:nref], w, None, 2, 1))).detach()
    assert lib.cn_set_tuning(28, 2 if B < 32 else 1) == 0
    try:
        for out_plain in (False, True):
            got = {}
            for key33 in (1, 0):
                assert lib.cn_set_tuning(33, key33) == 0
                pb = PlanBuilder(dev, B, H, W, split=True)
                y = pb.conv(pb.packed(_nhwc_act(x, dev)), w, bn=bn, relu=True, stride=2, padding=1,
                            out_plain=out_plain)
                assert y.fmt == ("f32" if out_plain else "f32s") and (y.H, y.W) == tuple(ref.shape[2:])
                _run(pb)
                got[key33] = y.to_float().clone()
                if key33 == 1:
                    raw = y.t.clone()
                    _run(pb)
                    assert torch.equal(raw, y.t), "not deterministic run to run"
            _check(got[1][:nref].permute(0, 3, 1, 2).cpu(), ref)
            d = float((got[1] - got[0]).abs().max()) / max(1.0, float(got[0].abs().max()))
            assert d < 2e-5, d
    finally:
        lib.cn_set_tuning(28, 1)
        lib.cn_set_tuning(33, 1)


@pytest.mark.parametrize("form", ["conv", "conv_res", "conv_plain_res", "deconv", "s2", "heads"])
def test_persistent_kernel_pipelined_schedule_is_bit_identical(dev, form):
    """The pipelined fragment schedule of the persistent kernel (cn_conv3x3p.hip, template PIPE: key 30
    bit 2, the default) sums every output in the order of the unpipelined one: the raw output tensors
    are BIT-identical, launch after launch, in every form -- 3x3 / s1 without / with an f32s residual,
    plain output with a residual, the transposed convolution, stride 2, the fused heads -- on forced
    small shapes with edge tiles and at the benchmark batch (where two workgroups share a CU and a
    fragment lost to an early DMA refill or a late LDS read would show as a rare differing tile)."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    lib = native.lib()
    shapes = {"conv": [(2, 96, 20, 24, 96), (32, 64, 128, 128, 64), (32, 256, 32, 32, 256)],
              "conv_res": [(3, 64, 17, 40, 64), (32, 64, 128, 128, 64), (32, 512, 16, 16, 512)],
              "conv_plain_res": [(2, 128, 16, 32, 128), (32, 128, 64, 64, 128)],
              "deconv": [(2, 128, 37, 45, 96), (32, 128, 64, 64, 64)],
              "s2": [(1, 96, 17, 23, 96), (32, 64, 128, 128, 128)],
              "heads": [(1, 64, 20, 24, 0), (32, 64, 128, 128, 0)]}[form]
    try:
        for (B, Cin, H, W, Cout) in shapes:
            assert lib.cn_set_tuning(28, 2 if B < 32 else 1) == 0
            x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1)).relu_()
            xa = _nhwc_act(x, dev)
            res = _nhwc_act(torch.from_numpy(synth.normal((B, Cout, H, W), 1.0, 5)), dev) if "res" in form else None
            pairs = _heads_case(Cin, 64, {"hm": 80, "wh": 2, "reg": 2}, seed=B) if form == "heads" else None
            raws = {}
            for knobs in (0, 2):
                assert lib.cn_set_tuning(30, knobs) == 0
                pb = PlanBuilder(dev, B, H, W, split=True)
                xin = pb.packed(xa)
                if form == "heads":
                    outs = pb.heads_from_convs(xin, pairs)
                    ts = [outs[n].t for n in ("hm", "wh", "reg")]
                elif form == "deconv":
                    w = torch.from_numpy(synth.normal((Cin, Cout, 4, 4), (2.0 / (Cin * 4)) ** 0.5, 2))
                    ts = [pb.conv_transpose4x4s2(xin, w, bn=_bn(Cout, 3), relu=True).t]
                else:
                    w = torch.from_numpy(synth.normal((Cout, Cin, 3, 3), (2.0 / (Cin * 9)) ** 0.5, 2))
                    r = None
                    if res is not None:
                        r = pb.packed(res) if form == "conv_res" else pb.plain(res)
                    ts = [pb.conv(xin, w, bn=_bn(Cout, 3), relu=True, stride=2 if form == "s2" else 1, padding=1,
                                  residual=r, out_plain=(form == "conv_plain_res")).t]
                reps = 6 if B == 32 else 2
                for rep in range(reps):
                    for t in ts:
                        t.zero_()
                    _run(pb)
                    if knobs == 0 and rep == 0:
                        raws = [t.clone() for t in ts]
                        assert all(float(t.abs().max()) > 0 for t in raws)
                    for t, r0 in zip(ts, raws):
                        assert torch.equal(t, r0), (form, (B, Cin, H, W, Cout), knobs, rep)
    finally:
        lib.cn_set_tuning(28, 1)
        lib.cn_set_tuning(30, 2)


@pytest.mark.parametrize("persistent", [1, 0])
def test_concat_members_written_in_place(dev, persistent):
    """Root.forward's torch.cat (pose_dla_dcn.py:157-165) without copies: the concatenation buffer is
    allocated first, a max-pool (f32s in, f32s out at the buffer's pitch: cn_maxpool_nhwc_f32s), a 3x3
    block output with a residual at ANOTHER pixel pitch (cn_conv_desc.res_pitch) and a 3x3 layer whose
    residual is its neighbour slice write their members in place; a fourth member is copied.  Against
    torch on the CPU, and the launch list holds exactly one copy."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    B, C, H, W = 8, 64, 64, 64      # (grids large enough that no layer wants split-K)
    native.lib().cn_set_tuning(28, persistent)     # both 3x3 kernels take a residual pitch
    x = torch.from_numpy(synth.normal((B, C, 2 * H, 2 * W), 1.0, 1)).relu_()
    r = torch.from_numpy(synth.normal((B, C, H, W), 1.0, 2))
    extra = torch.from_numpy(synth.normal((B, 32, H, W), 1.0, 3)).relu_()
    w1 = torch.from_numpy(synth.normal((C, C, 3, 3), (2.0 / (C * 9)) ** 0.5, 4))
    w2 = torch.from_numpy(synth.normal((C, C, 3, 3), (2.0 / (C * 9)) ** 0.5, 5))
    bn1, bn2 = _bn(C, 6), _bn(C, 7)
    bottom = F.max_pool2d(x, 2, 2)
    x1 = F.relu(bn1(F.conv2d(bottom, w1, None, 1, 1)) + r)
    x2 = F.relu(bn2(F.conv2d(x1, w2, None, 1, 1)) + x1)
    ref = torch.cat([x2, x1, extra, bottom], 1).detach()
    pb = PlanBuilder(dev, B, 2 * H, 2 * W, split=True)
    xa = pb.packed(_nhwc_act(x, dev))
    ra = pb.packed(_nhwc_act(r, dev))
    ea = pb.packed(_nhwc_act(extra, dev))
    n0 = len(pb.ops)
    buf, slots = pb.concat_buffer(B, H, W, [C, C, 32, C])
    assert buf is not None and buf.pitch == 3 * C + 32
    pooled = pb.maxpool(xa, 2, 2, 0, out=slots[3])
    a1 = pb.conv(pooled, w1, bn=bn1, relu=True, residual=ra, padding=1, out=slots[1])       # residual pitch 64, output pitch 224
    a2 = pb.conv(a1, w2, bn=bn2, relu=True, residual=a1, padding=1, out=slots[0])
    assert all(a.t is buf.t for a in (pooled, a1, a2))
    cat = pb.concat([a2, a1, ea, pooled], into=buf)
    kinds = [k for k, _ in pb.trace[n0:]]
    assert kinds.count("copy") == 1 and kinds.count("maxpool") == 1 and kinds.count("convert") == 0, kinds
    try:
        _run(pb)
    finally:
        native.lib().cn_set_tuning(28, 1)
    _check(cat.to_float().permute(0, 3, 1, 2).cpu(), ref)


def test_heads_fused_nchw_outputs(dev):
    from centernet_amd.engine import PlanBuilder
    B, F_, H, W = 2, 64, 32, 32
    heads = {"hm": 80, "wh": 2, "reg": 2}
    mods = {}
    for i, (h, c) in enumerate(heads.items()):
        seq = torch.nn.Sequential(torch.nn.Conv2d(F_, 64, 3, padding=1), torch.nn.ReLU(),
                                  torch.nn.Conv2d(64, c, 1))
        synth.fill_state_dict_(seq, 10 + i)
        mods[h] = seq.eval()
    x = torch.from_numpy(synth.normal((B, F_, H, W), 1.0, 1))
    pb = PlanBuilder(dev, B, H, W)
    outs = pb.heads(_nhwc_act(x, dev), mods)
    _run(pb)
    for h in heads:
        _check(outs[h].t.cpu(), mods[h](x).detach())


@pytest.mark.parametrize("cfg", [
    (2, 64, 32, 32, 64, 3, 1, 1, True, True), (2, 256, 16, 16, 384, 3, 2, 1, True, False),
    (1, 128, 17, 19, 256, 1, 1, 0, False, True), (1, 384, 8, 8, 384, 3, 1, 1, True, True),
    (1, 512, 4, 4, 512, 3, 1, 1, True, True),
])
def test_fp16_conv_vs_fp32_reference(dev, cfg):
    """fp16 operands / fp32 accumulate (configs[4]): against torch fp32 on the SAME
    fp16-rounded inputs and weights, so only accumulation order and the final fp16 rounding
    differ: |diff| <= 2e-3 * (1 + |ref|)."""
    from centernet_amd.engine import PlanBuilder, Act
    B, Cin, H, W, Cout, k, s, p, relu, use_res = cfg
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 1)).half()
    w = torch.from_numpy(synth.normal((Cout, Cin, k, k), (2.0 / (Cin * k * k)) ** 0.5, 2)).half()
    bn = _bn(Cout, 4)
    ref = bn(F.conv2d(x.float(), w.float(), None, s, p))
    res = None
    if use_res:
        res = torch.from_numpy(synth.normal(tuple(ref.shape), 1.0, 5)).half()
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    pb = PlanBuilder(dev, B, H, W, dtype=torch.float16)
    xa = Act(x.permute(0, 2, 3, 1).contiguous().to(dev), B, H, W, Cin)
    ra = Act(res.permute(0, 2, 3, 1).contiguous().to(dev), B, ref.shape[2], ref.shape[3], Cout) if use_res else None
    y = pb.conv(xa, w.float(), bn=bn, relu=relu, residual=ra, stride=s, padding=p)
    _run(pb)
    got = y.t.permute(0, 3, 1, 2).float().cpu()
    err = (got - ref.detach()).abs() / (1 + ref.detach().abs())
    assert float(err.max()) < 2e-3, float(err.max())


def test_fp16_stem_and_nchw_head(dev):
    from centernet_amd.engine import PlanBuilder, Act
    B, H, W = 1, 64, 64
    x = synth.images(B, H, W, 3)
    w = torch.from_numpy(synth.normal((128, 3, 7, 7), (2.0 / 147) ** 0.5, 2))
    bn = _bn(128, 4)
    ref = F.relu(bn(F.conv2d(x.half().float(), w.half().float(), None, 2, 3))).detach()
    pb = PlanBuilder(dev, B, H, W, dtype=torch.float16)
    xin = pb.set_input(3)
    y = pb.conv(xin, w, bn=bn, relu=True, stride=2, padding=3)
    w2 = torch.from_numpy(synth.normal((80, 128, 1, 1), 0.1, 7))
    b2 = torch.from_numpy(synth.normal((80,), 0.1, 8))
    z = pb.conv(y, w2, bias=b2, out_nchw=True)
    pb.input.t = x.to(dev)
    _run(pb)
    got = y.t.permute(0, 3, 1, 2).float().cpu()
    err = (got - ref).abs() / (1 + ref.abs())
    assert float(err.max()) < 3e-3, float(err.max())
    assert z.t.dtype == torch.float32
    ref2 = F.conv2d(got.half().float(), w2.half().float(), b2)
    err2 = (z.t.cpu() - ref2).abs() / (1 + ref2.abs())
    assert float(err2.max()) < 2e-3, float(err2.max())


@pytest.mark.parametrize("cfg", [
    # B, F, H, W, {head: classes}
    (2, 64, 32, 32, {"hm": 80, "wh": 2, "reg": 2}),
    (1, 64, 21, 45, {"hm": 80, "wh": 2, "reg": 2}),                 # ragged tiles, W >= 32
    (2, 64, 12, 20, {"hm": 3, "dep": 1}),                            # W < 32: 8x16 tiles
    (1, 256, 16, 16, {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}),
    (1, 64, 8, 40, {"big": 130, "wh": 2}),                           # > 96 channels: two passes
    # head_conv = 256 (pose_dla_dcn.py:456-468, large_hourglass.py): four 64-channel slices of the
    # hidden layer, their 1x1 products accumulated in registers
    (2, 64, 32, 32, {"hm": 80, "wh": 2, "reg": 2}, 256),
    (1, 64, 19, 33, {"hm": 1, "wh": 2, "hps": 34, "reg": 2, "hm_hp": 17, "hp_offset": 2}, 256),
    (1, 256, 12, 20, {"hm": 80, "wh": 2, "reg": 2}, 256),            # hourglass: 256-channel features
    (1, 96, 16, 16, {"hm": 5, "wh": 2}, 128),
])
@pytest.mark.parametrize("split", [True, False], ids=["f32s", "fp32mfma"])
def test_fused_heads(dev, cfg, split):
    """cn_heads3x3_1x1 (one launch for all heads, hidden channels kept in LDS) vs the
    per-head Sequential(conv3x3, ReLU, conv1x1) of resnet_dcn.py:155-177 on torch CPU."""
    from centernet_amd.engine import PlanBuilder
    hidden = cfg[5] if len(cfg) > 5 else 64
    B, Fc, H, W, heads = cfg[:5]
    x = torch.from_numpy(synth.normal((B, Fc, H, W), 1.0, 11))
    pairs, ref = {}, {}
    for i, (name, classes) in enumerate(heads.items()):
        c1 = torch.nn.Conv2d(Fc, hidden, 3, padding=1, bias=True)
        c2 = torch.nn.Conv2d(hidden, classes, 1, bias=True)
        with torch.no_grad():
            c1.weight.copy_(torch.from_numpy(synth.normal(tuple(c1.weight.shape), (2.0 / (Fc * 9)) ** 0.5, 20 + i)))
            c1.bias.copy_(torch.from_numpy(synth.normal((hidden,), 0.2, 30 + i)))
            c2.weight.copy_(torch.from_numpy(synth.normal(tuple(c2.weight.shape), 0.15, 40 + i)))
            c2.bias.copy_(torch.from_numpy(synth.normal((classes,), 0.5, 50 + i)))
            ref[name] = c2(F.relu(c1(x)))
        pairs[name] = (c1, c2)
    for packed_in in ((False, True) if split else (False,)):
        pb = PlanBuilder(dev, B, H, W, split=split)
        assert pb.fuse_heads
        xa = _nhwc_act(x, dev)
        if packed_in:
            xa = pb.packed(xa)
        n0 = len(pb.ops)
        outs = pb.heads_from_convs(xa, pairs)
        assert len(pb.ops) == n0 + 1, "heads with 64..256 hidden channels must be a single fused launch"
        _run(pb)
        for name in heads:
            assert outs[name].nchw and tuple(outs[name].t.shape) == tuple(ref[name].shape)
            _check(outs[name].t.cpu(), ref[name])


def _heads_case(Fc, hidden, heads, seed=0):
    pairs = {}
    for i, (name, classes) in enumerate(heads.items()):
        c1 = torch.nn.Conv2d(Fc, hidden, 3, padding=1, bias=True)
        c2 = torch.nn.Conv2d(hidden, classes, 1, bias=(name != "nobias"))
        with torch.no_grad():
            c1.weight.copy_(torch.from_numpy(synth.normal(tuple(c1.weight.shape), (2.0 / (Fc * 9)) ** 0.5, seed + 20 + i)))
            c1.bias.copy_(torch.from_numpy(synth.normal((hidden,), 0.2, seed + 30 + i)))
            c2.weight.copy_(torch.from_numpy(synth.normal(tuple(c2.weight.shape), 0.15, seed + 40 + i)))
            if c2.bias is not None:
                c2.bias.copy_(torch.from_numpy(synth.normal((classes,), 0.5, seed + 50 + i)))
        pairs[name] = (c1, c2)
    return pairs


@pytest.mark.parametrize("forced", [2, 1])
def test_fused_heads_on_the_persistent_kernel(dev, forced):
    """cn_heads3x3_1x1 with a 64-wide hidden layer on the persistent loader / consumer kernel
    (cn_conv3x3p.hip, HEADS: hidden layer in registers, 1x1 weights as packed fragments) against the
    per-head Sequential(conv3x3, ReLU, conv1x1) of resnet_dcn.py:155-177 on torch CPU and against
    the one-tile-per-workgroup kernel (cn_set_tuning key 31 = 0): edge tiles, Cin padding, one to
    five heads, 1 .. 96 outputs, a head without bias; key 28 = 2 forces the kernel onto small
    shapes, = 1 is the library's own routing (taken at the benchmark shape)."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    lib = native.lib()
    cases = [(2, 64, 32, 32, {"hm": 80, "wh": 2, "reg": 2}),
             (1, 64, 20, 24, {"hm": 80, "wh": 2, "reg": 2}),          # edge tiles in both directions
             (3, 96, 17, 40, {"hm": 33, "nobias": 96, "a": 1, "b": 64, "c": 5}),
             (2, 32, 8, 16, {"hm": 3})] if forced == 2 else \
            [(32, 64, 128, 128, {"hm": 80, "wh": 2, "reg": 2})]
    assert lib.cn_set_tuning(28, forced) == 0
    try:
        for (B, Fc, H, W, heads) in cases:
            pairs = _heads_case(Fc, 64, heads, seed=B + H)
            x = torch.from_numpy(synth.normal((B, Fc, H, W), 1.0, 11)).relu_()
            nref = min(B, 2)
            ref = {n: pairs[n][1](F.relu(pairs[n][0](x[:nref]))) for n in heads}
            got = {}
            for key31 in (1, 0):
                assert lib.cn_set_tuning(31, key31) == 0
                pb = PlanBuilder(dev, B, H, W, split=True)
                outs = pb.heads_from_convs(pb.packed(_nhwc_act(x, dev)), pairs)
                assert len(pb.ops) == 2, "convert + one fused launch"
                _run(pb)
                got[key31] = {n: outs[n].t.clone() for n in heads}
                if key31 == 1:      # run-to-run bit equality of the persistent kernel
                    _run(pb)
                    for n in heads:
                        assert torch.equal(outs[n].t, got[1][n]), n
            for n in heads:
                _check(got[1][n][:nref].cpu(), ref[n])
                # the two kernels sum in different orders: equal to fp32 rounding, every image
                d = float((got[1][n] - got[0][n]).abs().max()) / max(1.0, float(got[0][n].abs().max()))
                assert d < 2e-5, (n, d)
    finally:
        lib.cn_set_tuning(28, 1)
        lib.cn_set_tuning(31, 1)


@pytest.mark.parametrize("form", [0, 1, 2, 3])
def test_fused_heads_hidden_layer_forms(dev, form):
    """cn_set_tuning key 26: the f32s fused heads with the hidden layer staged through LDS (0), kept
    in registers for hidden widths > 64 (1, default), in 128-wide slices (2) and for 64-wide hidden
    layers too (3) -- all against the per-head Sequential on torch CPU, plus run-to-run bit equality
    at a batch that puts several workgroups on every CU."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    lib = native.lib()
    assert lib.cn_set_tuning(26, form) == 0
    try:
        for (B, Fc, H, W, heads, hidden) in [(2, 64, 32, 32, {"hm": 80, "wh": 2, "reg": 2}, 256),
                                              (1, 64, 19, 33, {"hm": 1, "hps": 34, "hm_hp": 17}, 256),
                                              (1, 96, 8, 64, {"hm": 5, "wh": 2}, 128),
                                              (2, 64, 16, 32, {"hm": 80, "wh": 2}, 64)]:
            x = torch.from_numpy(synth.normal((B, Fc, H, W), 1.0, 11))
            pairs, ref = {}, {}
            for i, (name, classes) in enumerate(heads.items()):
                c1 = torch.nn.Conv2d(Fc, hidden, 3, padding=1, bias=True)
                c2 = torch.nn.Conv2d(hidden, classes, 1, bias=True)
                with torch.no_grad():
                    c1.weight.copy_(torch.from_numpy(synth.normal(tuple(c1.weight.shape), (2.0 / (Fc * 9)) ** 0.5, 20 + i)))
                    c1.bias.copy_(torch.from_numpy(synth.normal((hidden,), 0.2, 30 + i)))
                    c2.weight.copy_(torch.from_numpy(synth.normal(tuple(c2.weight.shape), 0.15, 40 + i)))
                    c2.bias.copy_(torch.from_numpy(synth.normal((classes,), 0.5, 50 + i)))
                    ref[name] = c2(F.relu(c1(x)))
                pairs[name] = (c1, c2)
            pb = PlanBuilder(dev, B, H, W, split=True)
            outs = pb.heads_from_convs(pb.packed(_nhwc_act(x, dev)), pairs)
            _run(pb)
            for name in heads:
                _check(outs[name].t.cpu(), ref[name])
        # determinism at B = 32, 64 x 64 maps (2048 tiles x heads: several workgroups per CU)
        B, Fc, H, W, hidden = 32, 64, 64, 64, 256
        pairs = {}
        for i, (name, classes) in enumerate({"hm": 80, "wh": 2}.items()):
            c1 = torch.nn.Conv2d(Fc, hidden, 3, padding=1, bias=True)
            c2 = torch.nn.Conv2d(hidden, classes, 1, bias=True)
            pairs[name] = (c1, c2)
        pb = PlanBuilder(dev, B, H, W, split=True)
        xa = pb.packed(_nhwc_act(torch.randn((B, Fc, H, W)).relu_(), dev))
        outs = pb.heads_from_convs(xa, pairs)
        first = None
        for _ in range(6):
            _run(pb)
            cur = torch.cat([outs[n].t.reshape(-1) for n in outs]).clone().view(torch.int32)
            if first is None:
                first = cur
            assert torch.equal(cur, first)
    finally:
        lib.cn_set_tuning(26, 1)


def test_f32s_kernels_are_run_to_run_deterministic(dev):
    """Repeated launches at the benchmark batch (several workgroups per CU) give bit-identical
    results.  Guards the operand hazard found in round 2: a fragment read scheduled behind an
    MFMA into that MFMA's source registers can overwrite them before a queued MFMA reads them
    (csrc/cn_conv.hip, SPLIT compute)."""
    from centernet_amd.dcn_v2 import DCN
    from centernet_amd.engine import PlanBuilder, Act
    B = 32
    cases = []
    x64 = Act(torch.randn((B, 64, 64, 64), device=dev).relu_(), B, 64, 64, 64)
    pb = PlanBuilder(dev, B, 64, 64, split=True)
    cases.append((pb, pb.conv(x64, torch.randn((64, 64, 3, 3)) * 0.05, relu=True, stride=1, padding=1)))
    pb = PlanBuilder(dev, B, 64, 64, split=True)
    cases.append((pb, pb.conv(x64, torch.randn((128, 64, 3, 3)) * 0.05, relu=True, stride=2, padding=1)))
    for (C, H, Co) in [(256, 32, 128), (128, 64, 64)]:
        m = DCN(C, Co, (3, 3), 1, 1)
        synth.fill_state_dict_(m, 5)
        pb = PlanBuilder(dev, B, H, H, split=True)
        xa = Act(torch.randn((B, H, H, C), device=dev).relu_(), B, H, H, C)
        cases.append((pb, pb.dcn(xa, m, out_plain=True)))
    for pb, y in cases:
        first = None
        for _ in range(12):
            _run(pb)
            cur = y.t.clone().view(torch.int32)
            if first is None:
                first = cur
            assert torch.equal(cur, first)


@pytest.mark.parametrize("lds_weights", [0, 1])
def test_f32s_halo_weight_forms(dev, lds_weights):
    """128-wide f32s tiles: weights streamed into registers from the fragment-ordered copy
    (default) and the per-tap LDS weight tile (cn_set_tuning key 20) -- odd channel counts,
    several chunks, narrow and wide maps, residual."""
    from centernet_amd import native
    lib = native.lib()
    lib.cn_set_tuning(20, lds_weights)
    try:
        _conv_case(dev, (2, 256, 32, 32, 256, 3, 1, 1, False, True, True, True), split=True)
        _conv_case(dev, (3, 160, 19, 27, 130, 3, 1, 1, True, False, True, False), split=True)
        _conv_case(dev, (20, 96, 12, 20, 200, 3, 1, 1, True, True, False, False), split=True)
        _conv_case(dev, (16, 64, 64, 64, 192, 3, 1, 1, True, False, True, False), split=True)
    finally:
        lib.cn_set_tuning(20, 1)


@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e3])
@pytest.mark.parametrize("cfg", [(2, 64, 16, 16, 27, False), (1, 128, 24, 32, 27, False), (2, 512, 16, 16, 27, False),
                                 (1, 32, 8, 16, 18, True), (3, 96, 16, 48, 32, False), (32, 128, 64, 64, 27, False)])
def test_offset_conv_kernel(dev, cfg, scale):
    """The `conv_offset_mask` launch of the deformable modules (DCNv2/dcn_v2.py:52-62: Conv2d(Cin, 27, 3, 1, 1)
    on the plain fp32 tensor the sampler reads, 27 channels at pitch 32, plain output) on its own kernel
    (csrc/cn_offconv.hip: LDS-DMA halo, split in registers, eight waves in two teams; K split on the deep
    small map) against torch -- and against the LDS-halo kernel it replaces (cn_set_tuning key 39 = 0) --
    at activation scales the f32s exponents must absorb; the input-side range word reports max |x'|."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder, Act, exponent_for
    B, Cin, H, W, Cout, relu = cfg
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 21)) * scale
    w = torch.from_numpy(synth.normal((Cout, Cin, 3, 3), (2.0 / (Cin * 9)) ** 0.5, 22))
    bias = torch.from_numpy(synth.normal((Cout,), 0.3, 23)) * scale
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 1)
    if relu:
        ref = F.relu(ref)
    lib = native.lib()
    outs = []
    for key39 in (1, 0):
        assert lib.cn_set_tuning(39, key39) == 0
        try:
            pb = PlanBuilder(dev, B, H, W, split=True, exps={"x": exponent_for(float(x.abs().max()))})
            xa = Act(x.permute(0, 2, 3, 1).contiguous().to(dev), B, H, W, Cin, exp=pb._exp("x"), lid="x")
            om = pb._new(B, H, W, Cout, pitch=32, lid="om")
            y = pb.conv(xa, w, bias=bias, relu=relu, stride=1, padding=1, out=om, lid="om")
            assert y.fmt == "f32"
            _run(pb)
        finally:
            lib.cn_set_tuning(39, 1)
        got = y.t[..., :Cout].permute(0, 3, 1, 2).double().cpu()
        err = ((got - ref).abs() / (scale + ref.abs())).max()
        assert float(err) < TOL, (key39, float(err))
        outs.append(got)
    assert float(((outs[0] - outs[1]).abs() / (scale + ref.abs())).max()) < TOL


@pytest.mark.parametrize("cfg", [
    # B, Cin, H, W, Cout, stride, bias, bn, relu
    (32, 64, 128, 128, 128, 2, False, True, False),     # resdcn_18 layer2.0.downsample at the benchmark batch
    (4, 256, 32, 32, 512, 2, False, True, False),       # layer4.0.downsample
    (1, 64, 17, 19, 128, 2, False, True, False),        # odd map: (H - 1) / 2 + 1 rows, ragged last tile
    (2, 128, 33, 31, 256, 2, True, False, True),
    (3, 96, 9, 11, 160, 1, True, False, True),          # stride 1, three chunks, two channel groups of the second block row
    (2, 64, 8, 8, 32, 1, False, True, False),           # one block of output channels
    (1, 512, 16, 16, 512, 1, False, True, True),        # deep K: the four-chunk ring wraps
])
def test_projection_1x1_kernel(dev, cfg):
    """1x1 convolutions on f32s tensors without a residual operand -- the `downsample` projections
    (resnet_dcn.py:179-195, applied :52-56) -- on the direct-fragment kernel (csrc/cn_proj.hip): f32s and plain
    output, against torch fp64, and against the implicit-GEMM kernel it replaces (cn_set_tuning key 46 = 0)."""
    from centernet_amd import native
    from centernet_amd.engine import PlanBuilder
    B, Cin, H, W, Cout, s, use_bias, use_bn, relu = cfg
    x = torch.from_numpy(synth.normal((B, Cin, H, W), 1.0, 31))
    w = torch.from_numpy(synth.normal((Cout, Cin, 1, 1), (2.0 / Cin) ** 0.5, 32))
    bias = torch.from_numpy(synth.normal((Cout,), 0.3, 33)) if use_bias else None
    bn = _bn(Cout, 34) if use_bn else None
    ref = F.conv2d(x.double(), w.double(), bias.double() if use_bias else None, s, 0)
    if bn is not None:
        ref = bn.double()(ref)
        bn.float()
    if relu:
        ref = F.relu(ref)
    ref = ref.detach()
    lib = native.lib()
    got = {}
    try:
        for key46 in (1, 0):
            assert lib.cn_set_tuning(46, key46) == 0
            for out_plain in (False, True):
                pb = PlanBuilder(dev, B, H, W, split=True)
                xs = pb.packed(_nhwc_act(x, dev))
                y = pb.conv(xs, w, bias=bias, bn=bn, relu=relu, stride=s, padding=0, out_plain=out_plain)
                assert y.fmt == ("f32" if out_plain else "f32s")
                yp = pb.plain(y)
                _run(pb)
                got[key46, out_plain] = yp.t[..., :Cout].permute(0, 3, 1, 2).cpu().double()
                err = (got[key46, out_plain] - ref).abs() / (1 + ref.abs())
                assert float(err.max()) < TOL, (key46, out_plain, float(err.max()))
    finally:
        lib.cn_set_tuning(46, 1)
    # the two kernels agree far inside the bar (same arithmetic, different summation order)
    assert float((got[1, True] - got[0, True]).abs().max()) < 2e-5 * float(ref.abs().max())
