"""bench.pmc_traffic: per-class HBM bytes per forward step from a committed rocprofv3 PMC summary
(CPU test: parses the newest committed profiles/r*_pmc_traffic_cfg1.json / cfg2.json)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load():
    import bench
    return bench


def test_pmc_traffic_classes_and_step_normalisation():
    bench = _load()
    got, name = bench.pmc_traffic("cfg1")
    assert name and name.endswith("pmc_traffic_cfg1.json")
    # resdcn_18: 27 conv-class launches per forward (+ the fp32 calibration pass's share, at most
    # two launches' worth after rounding), three deformable layers, ONE decode launch (round 4: the
    # heat-map read once -- the decode's bytes are the map plus a few MB of keys)
    conv_bytes, conv_n = got["conv"]
    dcn_bytes, dcn_n = got["dcn"]
    dec_bytes, dec_n = got["decode"]
    assert 27 <= conv_n <= 30 and dcn_n == 3 and dec_n == 1
    # orders of magnitude: a B = 32 step moves a few GB through the conv class, the heat-map once
    # or twice through the decode, and the deformable layers stay near their algorithmic bytes
    assert 3e9 < conv_bytes < 9e9
    assert 1.5e8 < dec_bytes < 4e8
    assert 2.0e8 < dcn_bytes < 4.5e8          # 3 x 67 MB algorithmic: <= 1.5x


def test_pmc_traffic_dla_counts_the_window_kernel_as_dcn():
    bench = _load()
    got, _ = bench.pmc_traffic("cfg2")
    assert got["dcn"][1] == 16                 # dla_34: 16 deformable layers per forward
    assert "conv" in got and "decode" in got


def test_pmc_traffic_unknown_config_is_empty():
    bench = _load()
    assert bench.pmc_traffic("cfg99") == ({}, None)
