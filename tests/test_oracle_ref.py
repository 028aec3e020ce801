"""oracle/_ref -- the REFERENCE's own native code built test-only (oracle/Makefile `_ref`) --
against our restatements, bit for bit.

  * DCNv2: modulated_deformable_im2col_gpu_kernel + dmcn_im2col_bilinear
    (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47,118-180) compiled for the host vs
    oracle/dcn_v2_oracle.c: columns identical on the resdcn_18 / dla_34 layer list,
    deformable groups, strides / pads / dilations, stress offsets, window edges.
  * soft-NMS: external/nms.pyx:77-275 (cython) vs oracle/post_oracle.soft_nms and the product's
    cn_soft_nms_f32: kept count and the whole in-place array.

Skipped when oracle/_ref was never built (needs /root/reference at build time); the committed
fixtures tests/golden/ref_golden.npz (made from these libraries) keep the pin alive in that case
(test_ref_fixtures_* below and tests/test_gpu_dcn.py).
"""
import os

import numpy as np
import pytest

from oracle import cref, post_oracle, ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")

# (Cin, H, W) of every distinct DCN layer input in resdcn_18 (resnet_dcn.py:149-153,221) and
# dla_34 (pose_dla_dcn.py:360-413,437-443)
LAYERS = [(512, 16, 16), (256, 32, 32), (128, 64, 64), (64, 128, 128)]


def _inputs(Cin, H, W, seed, kh=3, kw=3, stride=1, pad=1, dil=1, dg=1, sigma=2.0, stress=False):
    rs = np.random.RandomState(seed)
    Ho, Wo = cref.out_hw(H, W, kh, kw, stride, pad, dil)
    x = rs.standard_normal((Cin, H, W)).astype(np.float32)
    if stress:
        off = rs.uniform(-H, H, (dg * 2 * kh * kw, Ho, Wo)).astype(np.float32)
    else:
        off = (rs.standard_normal((dg * 2 * kh * kw, Ho, Wo)) * sigma).astype(np.float32)
    mask = (1.0 / (1.0 + np.exp(-rs.standard_normal((dg * kh * kw, Ho, Wo))))).astype(np.float32)
    return x, off, mask


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@needs_ref
@pytest.mark.parametrize("Cin,H,W", LAYERS)
@pytest.mark.parametrize("stress", [False, True])
def test_im2col_columns_bit_identical_on_layer_list(Cin, H, W, stress):
    x, off, mask = _inputs(Cin, H, W, seed=Cin + H + int(stress), stress=stress)
    a = ref.dcn_v2_im2col(x, off, mask)
    b = cref.dcn_v2_im2col(x, off, mask)
    assert np.array_equal(_bits(a), _bits(b))
    assert np.abs(a).max() > 0


@needs_ref
@pytest.mark.parametrize("cfg", [
    dict(Cin=64, H=24, W=20, dg=2),                         # DCNv2/test.py:169-179 example family
    dict(Cin=12, H=15, W=17, dg=3),
    dict(Cin=2, H=4, W=4),                                  # DCNv2/test.py:16-19
    dict(Cin=6, H=17, W=13, stride=2),
    dict(Cin=6, H=17, W=13, stride=2, pad=0),
    dict(Cin=6, H=19, W=16, dil=2, pad=2),
    dict(Cin=4, H=14, W=18, kh=5, kw=5, pad=2, dg=2),
    dict(Cin=4, H=9, W=9, kh=1, kw=1, pad=0),
    dict(Cin=8, H=11, W=12, kh=3, kw=1, pad=0, stride=1),
])
@pytest.mark.parametrize("stress", [False, True])
def test_im2col_general_domain(cfg, stress):
    cfg = dict(cfg)
    Cin, H, W = cfg.pop("Cin"), cfg.pop("H"), cfg.pop("W")
    x, off, mask = _inputs(Cin, H, W, seed=7, stress=stress, **cfg)
    a = ref.dcn_v2_im2col(x, off, mask, **cfg)
    b = cref.dcn_v2_im2col(x, off, mask, **cfg)
    assert np.array_equal(_bits(a), _bits(b))


@needs_ref
def test_im2col_window_edges_and_exact_integers():
    """Offsets landing exactly on -1, 0, H-1, H and on half-integers (the window rule,
    dcn_v2_im2col_cuda.cu:165, and the per-corner rule, :30-41)."""
    Cin, H, W = 3, 6, 7
    rs = np.random.RandomState(3)
    x = rs.standard_normal((Cin, H, W)).astype(np.float32)
    vals = np.array([-7, -2, -1.5, -1, -0.5, 0, 0.5, 1, H - 1, H - 0.5, H, W, W + 0.5], np.float32)
    off = rs.choice(vals, size=(18, H, W)).astype(np.float32)
    mask = np.ones((9, H, W), np.float32)
    assert np.array_equal(_bits(ref.dcn_v2_im2col(x, off, mask)), _bits(cref.dcn_v2_im2col(x, off, mask)))


@needs_ref
@pytest.mark.parametrize("dg,stride", [(1, 1), (2, 1), (1, 2)])
def test_forward_bit_identical(dg, stride):
    rs = np.random.RandomState(11 + dg)
    B, Cin, H, W, Cout = 2, 8 * dg, 13, 12, 6
    Ho, Wo = cref.out_hw(H, W, 3, 3, stride, 1, 1)
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    off = (rs.standard_normal((B, dg * 18, Ho, Wo)) * 2).astype(np.float32)
    mask = rs.uniform(0, 1, (B, dg * 9, Ho, Wo)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, 3, 3)) * 0.1).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    ya = ref.dcn_v2_forward(x, off, mask, w, b, stride=stride, dg=dg)
    yb = cref.dcn_v2_forward(x, off, mask, w, b, stride=stride, dg=dg)
    assert np.array_equal(_bits(ya), _bits(yb))


@needs_ref
def test_reference_kat_on_reference_kernel():
    """DCNv2/test.py:32-65 (zero offsets, mask 0.5, identity centre tap => 2*y == x) run on the
    reference's own kernel."""
    N, inC, inH, inW = 2, 2, 4, 4
    x = np.random.RandomState(0).standard_normal((N, inC, inH, inW)).astype(np.float32)
    w = np.zeros((inC, inC, 3, 3), np.float32)
    w[np.arange(inC), np.arange(inC), 1, 1] = 1.0
    y = ref.dcn_v2_forward(x, np.zeros((N, 18, inH, inW), np.float32),
                           np.full((N, 9, inH, inW), 0.5, np.float32), w, np.zeros(inC, np.float32))
    assert np.abs(2 * y - x).max() < 1e-10


def _nms_case(n, ncol, seed, crowded=True):
    rs = np.random.RandomState(seed)
    span = 120 if crowded else 2000
    xy = rs.uniform(0, span, (n, 2))
    wh = rs.uniform(10, 80, (n, 2))
    boxes = np.zeros((n, ncol), np.float32)
    boxes[:, 0:2] = xy
    boxes[:, 2:4] = xy + wh
    boxes[:, 4] = rs.uniform(0.0005, 1, n)
    if ncol > 5:
        boxes[:, 5:] = rs.uniform(0, 200, (n, ncol - 5))
    return boxes


@needs_ref
@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("ncol", [5, 39])
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_soft_nms_restatement_and_product_match_reference_cython(method, ncol, seed):
    """external/nms.pyx:77-275.  Crowded boxes + a threshold that discards many rows, so the
    swap-with-last bookkeeping (and soft_nms_39's column swap, :260-268) is exercised."""
    from centernet_amd.soft_nms import soft_nms, soft_nms_39
    base = _nms_case(80, ncol, seed * 13 + method)
    kw = dict(Nt=0.5, method=method, threshold=0.05 if seed else 0.001)
    r = base.copy()
    keep_r = (ref.soft_nms if ncol == 5 else ref.soft_nms_39)(r, **kw)
    o = base.copy()
    keep_o = post_oracle.soft_nms(o, **kw)
    p = base.copy()
    keep_p = (soft_nms if ncol == 5 else soft_nms_39)(p, **kw)
    assert keep_r == keep_o == keep_p
    if seed or method == 0:
        assert len(keep_r) < 80          # rows were discarded: the swap bookkeeping ran
    assert np.array_equal(_bits(r), _bits(o))
    assert np.array_equal(_bits(r), _bits(p))


@needs_ref
def test_soft_nms_edge_cases_match_reference_cython():
    from centernet_amd.soft_nms import soft_nms
    for boxes in (np.zeros((0, 5), np.float32),
                  np.array([[0, 0, 9, 9, 0.9]], np.float32),
                  np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [100, 100, 109, 109, 0.7]], np.float32),
                  np.array([[5, 5, 1, 1, 0.5], [0, 0, 9, 9, 0.5], [0, 0, 9, 9, 0.5]], np.float32)):
        for method in (0, 1, 2):
            r, p = boxes.copy(), boxes.copy()
            assert ref.soft_nms(r, Nt=0.5, method=method) == soft_nms(p, Nt=0.5, method=method)
            assert np.array_equal(_bits(r), _bits(p))


# ---- committed fixtures made from oracle/_ref (tests/golden/gen_golden_ref.py) ---------------

def test_ref_fixtures_pin_the_dcn_oracle():
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(GOLDEN, "gen_golden_ref.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for name, cfg in gen.DCN_CASES.items():
        x, off, mask, w, b, kw = gen.dcn_inputs(cfg)
        y = cref.dcn_v2_forward(x, off, mask, w, b, **kw)
        assert np.array_equal(_bits(y), _bits(z["dcn_" + name + "_y"])), name
        cols = cref.dcn_v2_im2col(x[0], off[0], mask[0], kh=w.shape[2], kw=w.shape[3], **kw)
        assert np.array_equal(_bits(cols.reshape(-1)[::gen.COL_STRIDE]), _bits(z["dcn_" + name + "_cols"])), name


def test_ref_fixtures_pin_soft_nms():
    from centernet_amd.soft_nms import soft_nms, soft_nms_39
    z = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_ref", os.path.join(GOLDEN, "gen_golden_ref.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for name, (ncol, seed, kw) in gen.NMS_CASES.items():
        base = gen.nms_inputs(ncol, seed)
        o, p = base.copy(), base.copy()
        keep_o = post_oracle.soft_nms(o, **kw)
        keep_p = (soft_nms if ncol == 5 else soft_nms_39)(p, **kw)
        assert len(keep_o) == len(keep_p) == int(z["nms_" + name + "_keep"])
        assert np.array_equal(_bits(o), _bits(z["nms_" + name + "_boxes"])), name
        assert np.array_equal(_bits(p), _bits(z["nms_" + name + "_boxes"])), name
