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
    got, name = bench.pmc_traffic("cfg1", check_head=False)   # the parser, whatever tree the profile is from
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
    got, _ = bench.pmc_traffic("cfg2", check_head=False)
    assert got["dcn"][1] == 16                 # dla_34: 16 deformable layers per forward
    assert "conv" in got and "decode" in got


def test_pmc_traffic_unknown_config_is_empty():
    bench = _load()
    assert bench.pmc_traffic("cfg99") == ({}, None)


def test_a_profile_taken_on_other_kernel_sources_is_refused(tmp_path, monkeypatch):
    """the PMC set must belong to the tree that runs.  A profile without a recorded head, or
    with another tree's, gives {"_stale": ...} (bench.py then prints traffic null and says why); the same
    file stamped with this tree's kernel_tree_sha is accepted."""
    import glob
    import json
    import shutil
    bench = _load()
    src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_cfg1.json")))[-1]
    os.makedirs(tmp_path / "profiles")
    for sub in ("centernet_amd/csrc", "include"):
        os.makedirs(tmp_path / sub)
    for f in glob.glob(os.path.join(ROOT, "centernet_amd", "csrc", "*.h*")):
        shutil.copy(f, tmp_path / "centernet_amd" / "csrc")
    shutil.copy(os.path.join(ROOT, "include", "centernet_amd.h"), tmp_path / "include")
    shutil.copy(os.path.join(ROOT, "centernet_amd", "engine.py"), tmp_path / "centernet_amd")
    raw = json.load(open(src))
    raw.pop("_meta", None)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    json.dump(raw, open(tmp_path / "profiles" / "r99_pmc_traffic_cfg1.json", "w"))
    got, name = bench.pmc_traffic("cfg1")
    assert name == "r99_pmc_traffic_cfg1.json" and list(got) == ["_stale"] and got["_stale"][0] is None
    raw["_meta"] = {"head": "0123456789abcdef"}
    json.dump(raw, open(tmp_path / "profiles" / "r99_pmc_traffic_cfg1.json", "w"))
    assert bench.pmc_traffic("cfg1")[0]["_stale"] == ("0123456789abcdef", bench.kernel_tree_sha())
    raw["_meta"] = {"head": bench.kernel_tree_sha()}
    json.dump(raw, open(tmp_path / "profiles" / "r99_pmc_traffic_cfg1.json", "w"))
    got, _ = bench.pmc_traffic("cfg1")
    assert "_stale" not in got and got["dcn"][1] == 3
    # one changed byte in a kernel source changes the head
    with open(tmp_path / "centernet_amd" / "csrc" / "cn_common.h", "a") as f:
        f.write("\n")
    assert "_stale" in bench.pmc_traffic("cfg1")[0]

