"""The compiler's gfx950 output of the kernels that place their own s_barrier, audited on the CPU.

Rule (the wrong tile of round 5): an LDS read is waited for before the barrier that
hands its source to the next writer.  `tools/audit_barriers.py` walks the basic blocks of the listing
with the LGKM queue as its state; this test holds that NO barrier of a shipped instantiation of the
persistent 3x3 kernel, the deformable team kernel and the offset convolution can be reached with a
ds_read in flight.  (The probe instantiation `conv3x3p_kernel<., ., DBG = true, ...>` is exempt: its
switchable paths are infeasible combinations to a path-insensitive walk, and nothing launches it
outside `tools/bench_c3p.py PROBE=1`.  The UNPIPELINED heads form, the A/B reference behind cn_set_tuning
key 30, is held at its three known reports: the 1x1 epilogue's last read of the output scale / bias
stash, whose wait sits inside the `pixel exists` branch; the stash is rewritten three steps into the
next item, each of which waits lgkmcnt(0) for its own fragments first -- LDS returns in order.)
"""
import importlib.util
import os
import subprocess
import shutil
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["cn_conv3x3p", "cn_dcn3", "cn_offconv"]


def _tool():
    spec = importlib.util.spec_from_file_location("audit_barriers", os.path.join(ROOT, "tools", "audit_barriers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _listing(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "centernet_amd", "csrc", name + ".hip")], check=True, capture_output=True)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_no_lds_read_in_flight_at_a_barrier(tmp_path):
    tool = _tool()
    with ThreadPoolExecutor(len(FILES)) as ex:
        listings = list(ex.map(lambda n: _listing(n, str(tmp_path)), FILES))
    seen = 0
    for path in listings:
        res = tool.audit(path)
        assert res, path
        for mangled, (nbar, flagged) in res.items():
            name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip() if shutil.which("c++filt") else mangled
            if "conv3x3p_kernel<" in name:
                args = name.split("conv3x3p_kernel<")[1].split(">")[0].split(",")
                if args[2].strip() == "true":  # DBG
                    continue
            seen += 1
            assert flagged == 0, "%s: %d of %d barriers reachable with an LDS read in flight" % (name, flagged, nbar)
    assert seen >= 20


def test_the_walk_sees_a_read_across_a_barrier(tmp_path):
    """the tool itself: a read that is waited for, one that is not, one behind a join"""
    tool = _tool()
    src = """
_Z1kv:
	ds_read_b128 v[0:3], v8
	s_waitcnt lgkmcnt(0)
	s_barrier
	ds_read_b128 v[4:7], v8
	s_cbranch_scc1 .LBB0_2
	s_waitcnt lgkmcnt(0)
.LBB0_2:
	s_barrier
	ds_write_b32 v8, v4
	s_waitcnt lgkmcnt(0)
	ds_read_b32 v9, v8
	ds_read_b32 v10, v8 offset:4
	s_waitcnt lgkmcnt(1)
	s_barrier
	s_endpgm
.Lfunc_end0:
"""
    p = tmp_path / "k.s"
    p.write_text(src)
    assert tool.audit(str(p)) == {"_Z1kv": (3, 2)}
