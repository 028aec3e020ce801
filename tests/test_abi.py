"""The C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/centernet_amd.h declares (no compute: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "centernet_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from centernet_amd import native
    path = native.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_status_strings_and_arch():
    from centernet_amd import native
    lib = native.lib()
    assert lib.cn_version() >= 100
    assert lib.cn_arch() == b"gfx950"
    assert lib.cn_status_string(0) == b"ok"
    assert b"workspace" in lib.cn_status_string(-3)


def test_device_code_is_gfx950_only():
    """The shared object carries exactly one offload target: gfx950."""
    from centernet_amd import native
    blob = open(native.build(), "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_shape_queries_without_gpu():
    """Pure host-side entry points work without a device."""
    from centernet_amd import native
    lib = native.lib()
    assert lib.cn_packed_conv_weight_floats(64, 64, 3, 3) == 9 * 64 * 64
    assert lib.cn_packed_conv_weight_floats(27, 512, 3, 3) == 9 * 32 * 512
    assert lib.cn_packed_conv_weight_floats(64, 3, 7, 7) == 64 * 160
    n = lib.cn_ctdet_decode_workspace_bytes(32, 80, 128, 128, 100)
    assert n >= 2 * 4 * 32 * 80 * 100
    assert lib.cn_ctdet_decode_workspace_bytes(1, 80, 128, 128, 100) > 0
    assert lib.cn_dcn_v2_forward_workspace_bytes(2, 64, 16, 16, 64, 3, 3, 0) > 0


def test_stem_maxpool_support_rule():
    """cn_stem_maxpool_supported: the shapes whose stem runs with the max-pool inside the kernel."""
    import ctypes
    from centernet_amd import native
    from centernet_amd.native import ConvDesc, LAYOUT_NCHW, LAYOUT_NHWC, DTYPE_F32
    lib = native.lib()

    def ok(B, H, W, Cout, k=7, s=2, cin=3, layout=LAYOUT_NCHW):
        d = ConvDesc(B=B, H=H, W=W, Cin=cin, Ho=(H + 6 - k) // s + 1, Wo=(W + 6 - k) // s + 1, Cout=Cout,
                     KH=k, KW=k, stride=s, pad_h=3, pad_w=3, dil=1, in_layout=layout,
                     out_layout=LAYOUT_NHWC, dtype=DTYPE_F32)
        return lib.cn_stem_maxpool_supported(ctypes.byref(d))
    assert ok(32, 512, 512, 64) == 1          # the headline configuration
    assert ok(8, 512, 512, 64) == 1           # strips of 4 pooled rows
    assert ok(1, 512, 512, 64) == 0           # too few strips to fill the chip: two launches
    assert ok(32, 512, 384, 64) == 0          # rows that are not whole 128-pixel tiles
    assert ok(32, 512, 512, 16) == 0          # DLA's 16-channel stem is a different kernel
    assert ok(32, 512, 512, 128) == 0
    assert ok(32, 512, 512, 64, s=1) == 0
    assert ok(32, 512, 512, 64, k=3) == 0
    assert ok(32, 512, 512, 64, cin=64, layout=LAYOUT_NHWC) == 0
    assert lib.cn_stem_maxpool_supported(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import pytest
    from centernet_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(native.NativeError):
        native.lib()


def test_product_path_never_imports_the_oracle_or_the_reference():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
    cpu_baseline may import it; nothing under centernet_amd/ may, nor may anything read
    /root/reference at run time."""
    import ast
    import glob
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "centernet_amd")
    offenders = []
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        src = open(path).read()
        tree = ast.parse(src)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                if n == "oracle" or n.startswith("oracle."):
                    offenders.append((path, n))
        if "/root/reference" in "".join(l for l in src.splitlines(True) if "open(" in l or "sys.path" in l):
            offenders.append((path, "/root/reference"))
    assert not offenders, offenders
