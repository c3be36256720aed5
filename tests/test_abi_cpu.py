"""The C-ABI library: loads, exports every symbol include/s3r.h declares, and its host-only entry
points (geometry, packing sizes, argument validation) behave — no GPU compute is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(s3r):
    import __graft_entry__ as g
    if not os.path.exists(s3r.LIB_PATH):
        g.build()
    return s3r.load_library()


def test_header_symbols_all_exported(s3r, lib):
    header = open(os.path.join(ROOT, "include", "s3r.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(s3r_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 17
    bound = set(s3r._lib.SIGNATURES)
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.s3r_abi_version() == 1


def test_struct_layouts_match_header(s3r):
    assert C.sizeof(s3r._lib.ConvDesc) == 12 * 4
    assert C.sizeof(s3r._lib.Layer) == 12 * 4 + 3 * 8
    assert C.sizeof(s3r._lib.ProfRecord) == 32


def _desc(s3r, layer, batch, n):
    return s3r._lib.make_desc(layer, batch, n)


def test_out_size_and_packed_elems(s3r, lib):
    spec = s3r.arch_spec
    for stage in ("encoder", "decoder", "point_head"):
        for l, n, m in spec.stage_table(stage):
            d = _desc(s3r, l, 2, n)
            assert lib.s3r_conv_out_size(C.byref(d)) == m
            e = C.c_int64(0)
            assert lib.s3r_conv_packed_elems(C.byref(d), C.byref(e)) == 0, lib.s3r_last_error()
            nd = spec.ndim(l)
            if l.name == "e1":
                assert e.value == 27 * 32
            elif l.name == "d4":
                assert e.value == l.cin
            elif l.op == "linear":
                assert e.value == l.cin * l.cout
            else:
                pad = (l.cout + 127) // 128 * 128
                assert e.value == l.k ** nd * l.cin * pad


def test_workspace_query(s3r, lib):
    spec = s3r.arch_spec
    rows = spec.stage_table("encoder")
    arr = (s3r._lib.Layer * len(rows))()
    for i, (l, n, m) in enumerate(rows):
        arr[i].desc = _desc(s3r, l, 4, n)
    need = lib.s3r_chain_workspace_elems(arr, len(rows))
    assert need == 4 * 64 * 112 * 112          # e2's output is the largest intermediate


def test_invalid_arguments_are_reported_not_crashed(s3r, lib):
    spec = s3r.arch_spec
    e = C.c_int64(0)
    bad = _desc(s3r, spec.Layer("x", "conv2d", 5, 8, 3, 1, 1), 1, 8)       # cin % 16 != 0
    assert lib.s3r_conv_packed_elems(C.byref(bad), C.byref(e)) == -1
    assert b"cin" in lib.s3r_last_error()
    bad = _desc(s3r, spec.Layer("x", "deconv3d", 16, 8, 3, 1, 1), 1, 8)    # unsupported transposed shape
    assert lib.s3r_conv_packed_elems(C.byref(bad), C.byref(e)) == -1
    bad = _desc(s3r, spec.Layer("x", "conv3d", 16, 8, 3, 1, 1), 0, 8)      # empty batch at the ABI
    assert lib.s3r_conv_packed_elems(C.byref(bad), C.byref(e)) == -1
    huge = _desc(s3r, spec.Layer("x", "conv3d", 64, 64, 3, 1, 1), 4096, 28)
    assert lib.s3r_conv_packed_elems(C.byref(huge), C.byref(e)) == -1
    assert b"split the batch" in lib.s3r_last_error()
    assert lib.s3r_cost_volume_forward(None, None, None, 1, 1, 1, 1, 1, None) == -1
    assert lib.s3r_chamfer_forward(None, None, None, None, None, None, 1, 1, 1, None) == -1
    assert lib.s3r_chain_forward(None, 0, None, None, None, None, 0, None) == -1


def test_missing_library_fails_loudly(s3r, monkeypatch, tmp_path):
    monkeypatch.setattr(s3r._lib, "_lib", None)
    monkeypatch.setattr(s3r._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(s3r.S3RError, match="no CPU fallback"):
        s3r._lib.load()
