"""Property tests (hypothesis) of the host-side restatements against the scalar oracles:
random sizes / matrices / detections, bit-exact agreement (1e-4 px on mapped coordinates).  CPU only, a few seconds."""
import numpy as np
from hypothesis import given, settings, strategies as st

from centernet_amd import image as I
from centernet_amd.post_process import ctdet_results_batch
from oracle import pre_oracle as P, post_oracle


@settings(max_examples=25, deadline=None)
@given(h=st.integers(1, 12), w=st.integers(1, 12), oh=st.integers(1, 10), ow=st.integers(1, 10),
       a=st.floats(-2, 2), b=st.floats(-2, 2), c=st.floats(-8, 8), d=st.floats(-2, 2),
       e=st.floats(-2, 2), f=st.floats(-8, 8), seed=st.integers(0, 99))
def test_host_warp_equals_oracle(h, w, oh, ow, a, b, c, d, e, f, seed):
    if abs(a * e - b * d) < 1e-3:
        return
    img = np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
    m = np.array([a, b, c, d, e, f]).reshape(2, 3)
    assert np.array_equal(I.warp_affine(img, m, (ow, oh)), P.cv_warp_affine_u8(img, m, (ow, oh)))


@settings(max_examples=25, deadline=None)
@given(h=st.integers(1, 10), w=st.integers(1, 10), oh=st.integers(1, 12), ow=st.integers(1, 12),
       seed=st.integers(0, 99))
def test_host_resize_equals_oracle(h, w, oh, ow, seed):
    img = np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
    assert np.array_equal(I.resize_bilinear(img, (ow, oh)), P.cv_resize_linear_u8(img, (ow, oh)))


@settings(max_examples=15, deadline=None)
@given(K=st.integers(1, 140), ncls=st.integers(1, 80), seed=st.integers(0, 999),
       cx=st.floats(10, 600), cy=st.floats(10, 600), s=st.floats(50, 900))
def test_batched_tail_equals_reference_loop(K, ncls, seed, cx, cy, s):
    rng = np.random.RandomState(seed)
    dets = np.zeros((2, K, 6), np.float32)
    dets[:, :, :4] = rng.uniform(-5, 133, (2, K, 4))
    dets[:, :, 4] = np.sort(rng.uniform(0, 1, (2, K)), axis=1)[:, ::-1]
    dets[:, :, 5] = rng.randint(0, ncls, (2, K))
    meta = {'c': np.array([cx, cy], np.float32), 's': float(s), 'out_height': 128, 'out_width': 128}
    got = ctdet_results_batch(dets.copy(), [meta, meta], ncls, scale=1, max_per_image=100)
    for i in range(2):
        ref = post_oracle.ctdet_results(dets[i:i + 1].copy(), meta, ncls, scale=1, max_per_image=100)
        for j in range(1, ncls + 1):
            # same rows, same order, same scores; coordinates within 1e-4 px: the product writes
            # the similarity in closed form, the oracle solves the reference's three-point
            # system, whose float32 construction points make it a similarity only up to their
            # rounding (exact for the half-integer centres real frames have; <= 3e-5 px here)
            assert got[i][j].shape == ref[j].shape and got[i][j].dtype == ref[j].dtype
            assert np.array_equal(got[i][j][:, 4], ref[j][:, 4])
            assert np.abs(got[i][j][:, :4] - ref[j][:, :4]).max(initial=0) <= 1e-4
