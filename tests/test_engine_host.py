"""Host-side logic of centernet_amd/engine.py that needs no device: f32s exponents, the bit-pattern
bounds of the range digest, which convolution forms take f32s arithmetic, weight-row prescaling."""
import os
import struct

import numpy as np
import pytest
import torch

from centernet_amd import engine
from centernet_amd.engine import PlanBuilder


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_range_digest_bounds_are_the_float_bit_patterns():
    assert engine.F16_MAX_BITS == _bits(engine.F16_MAX) == 0x477FE000
    assert engine.LOW_WATER_BITS == _bits(engine.LOW_WATER)
    # non-negative floats order like their bit patterns; inf and every NaN pattern lie above 65504
    vals = [0.0, 1e-30, engine.LOW_WATER, 1.0, 65503.9, 65504.0, 65505.0, float("inf")]
    assert [_bits(v) for v in vals] == sorted(_bits(v) for v in vals)
    assert _bits(float("inf")) > engine.F16_MAX_BITS and 0x7FC00000 > engine.F16_MAX_BITS


@pytest.mark.parametrize("absmax", [1e-6, 3e-3, 0.7, 1.0, 5.0, 511.9, 512.0, 1023.9, 1024.0, 6.5e4, 3e7])
def test_exponent_puts_the_largest_value_just_below_two_to_the_top(absmax):
    e = engine.exponent_for(absmax)
    stored = absmax * 2.0 ** -e
    assert 2.0 ** (engine.TOP_LOG2 - 1) <= stored < 2.0 ** engine.TOP_LOG2
    assert stored <= engine.F16_MAX                       # six binades of headroom above the top


def test_exponent_of_an_all_zero_tensor_is_zero_and_non_finite_calibration_is_refused():
    from centernet_amd import native
    for v in (0.0, -1.0):
        assert engine.exponent_for(v) == 0
    for v in (float("inf"), float("nan")):
        with pytest.raises(native.NativeError):
            engine.exponent_for(v)


def test_prescaled_rows_are_exact_powers_of_two_and_reach_the_weight_top():
    g = torch.Generator().manual_seed(3)
    w = torch.randn((7, 5, 3, 3), generator=g) * torch.tensor([1e-4, 1e-2, 1.0, 30.0, 2.0, 0.5, 0.0]).view(7, 1, 1, 1)
    wp, factor = engine.prescale_rows(w)
    assert wp.shape == w.shape and factor.shape == (7,)
    f = factor.double().numpy()
    assert np.all(np.log2(f) == np.round(np.log2(f)))            # powers of two: undone exactly
    assert torch.equal(wp * factor.view(7, 1, 1, 1), w)           # bit-exact round trip
    top = wp.abs().amax(dim=(1, 2, 3))
    nz = top > 0
    assert bool(((top[nz] >= 2.0 ** (engine.W_TOP_LOG2 - 1)) & (top[nz] < 2.0 ** engine.W_TOP_LOG2)).all())
    assert float(factor[6]) == 1.0 and float(top[6]) == 0.0       # an all-zero row is left alone


def test_which_convolutions_compute_in_f32s(monkeypatch):
    monkeypatch.delenv("CN_C16_F32S", raising=False)
    form = PlanBuilder._f32s_conv_form
    narrow = PlanBuilder._narrow_form
    # every trunk form
    assert form(3, 3, 1, 1, 1, False, 64, 64) and form(1, 1, 2, 0, 1, False, 64, 128)
    assert form(3, 3, 2, 1, 1, False, 64, 128) and form(3, 3, 1, 1, 1, True, 64, 80)
    # DLA's 16-channel layers: level0 (stride 1) in f32s arithmetic, level1 (stride 2) on the fp32 kernel
    assert narrow(3, 3, 1, 1, False, 16, 16) and narrow(3, 3, 1, 1, False, 16, 32)
    assert not narrow(3, 3, 1, 1, False, 16, 64) and not narrow(1, 1, 0, 1, False, 16, 16)
    assert form(3, 3, 1, 1, 1, False, 16, 16) and not form(3, 3, 2, 1, 1, False, 16, 32)
    monkeypatch.setenv("CN_C16_F32S", "0")
    assert not form(3, 3, 1, 1, 1, False, 16, 16)
