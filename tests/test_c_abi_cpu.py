"""include/s3r.h consumed from COMPILED C (tests/c_abi/consumer.c, plain gcc) — the boundary a cgo / JNI / N-API binding
goes through — and INTEGRATION.md's ctypes stub executed against the built library.  No GPU: planning entries only.

The boundary stands in for the reference's nn.Module forwards (/root/reference/README.md:91) and its one native binding,
extensions/chamfer_dist (README.md:64-65)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CABI = os.path.join(ROOT, "tests", "c_abi")
CSRC = os.path.join(ROOT, "stereo-3d-reconstruction_amd", "csrc")


def build_consumer(name, extra=()):
    """gcc -I include tests/c_abi/<name>.c against the in-tree libs3r_hip.so -> tests/c_abi/_build/<name>"""
    out_dir = os.path.join(CABI, "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, name)
    src = os.path.join(CABI, name + ".c")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", CSRC, "-ls3r_hip", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.fixture(scope="module")
def lib(s3r):
    import __graft_entry__ as g
    if not os.path.exists(s3r.LIB_PATH):
        g.build()
    return s3r.load_library()


@pytest.fixture(scope="module")
def consumer(lib):
    return build_consumer("consumer")


def desc_line(d):
    f = [getattr(d, n) for n, _ in type(d)._fields_]
    return "desc " + " ".join(str(int(v)) for v in f[:-1]) + f" {f[-1]!r}"


def chain_descs(s3r, stage, batch):
    rows = s3r.arch_spec.stage_table(stage)
    # (the stem e1 reads the renders unpadded — it predicates its own border — and a chain's first layer describes the caller's buffer)
    return [s3r._lib.make_desc(l, batch, n, tag=i, in_halo=0 if l.op == "linear" or l.name == "e1" else l.p)
            for i, (l, n, m) in enumerate(rows)]


def test_header_compiles_and_layout_is_abi8(consumer):
    """the _Static_asserts in consumer.c (sizeof 88 / 112 / 48 and every field offset) held at compile time; at run time the C side
    sees the same numbers the ctypes mirror declares"""
    out = subprocess.run([consumer], input="", capture_output=True, text=True, check=True).stdout
    kv = dict(t.split("=") for t in out.split()[1:])
    assert kv == {"abi": "8", "desc": "88", "layer": "112", "prof": "48", "algo_off": "72", "act_param_off": "84", "packed_w_off": "88"}


def test_ctypes_mirror_has_the_c_compilers_offsets(s3r):
    L = s3r._lib
    src = open(os.path.join(CABI, "consumer.c")).read()
    offs = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"OFF\((\w+), (\d+)\)", src))
    assert len(offs) == 22
    for name, _ in L.ConvDesc._fields_:
        assert getattr(L.ConvDesc, name).offset == offs[name], name
    assert (L.Layer.packed_w.offset, L.Layer.scale.offset, L.Layer.shift.offset) == (88, 96, 104)
    assert (L.ProfRecord.flops.offset, L.ProfRecord.exec_flops.offset, L.ProfRecord.algo.offset) == (16, 32, 40)


def test_planning_entries_from_c_equal_ctypes(s3r, lib, consumer):
    """e1 ... d4 and p1 ... p3: out size, packed size, scratch, Winograd input layout / size and the chains' workspaces, asked from
    compiled C, equal what the Python binding gets for the same descriptors"""
    lines, want = [], []
    for stage, batch in (("encoder", 4), ("decoder", 2), ("point_head", 2)):
        descs = chain_descs(s3r, stage, batch)
        for d in descs:
            lines.append(desc_line(d))
            e = C.c_int64(-1)
            rc = lib.s3r_conv_packed_elems(C.byref(d), C.byref(e))
            want.append(f"desc {lib.s3r_conv_out_size(C.byref(d))} {rc} {e.value} {lib.s3r_conv_scratch_elems(C.byref(d))} "
                        f"{lib.s3r_conv_wino_input_layout(C.byref(d))} {lib.s3r_conv_wino_input_elems(C.byref(d))}")
        if stage != "point_head":
            arr = (s3r._lib.Layer * len(descs))()
            for i, d in enumerate(descs):
                arr[i].desc = d
            lines.append(f"chain {len(descs)}")
            lines += [desc_line(d) for d in descs]
            want.append(f"chain {lib.s3r_chain_workspace_elems(arr, len(descs))}")
    for b, ci, co in ((2, 32768, 1024), (32, 1024, 6144)):
        lines.append(f"linear {b} {ci} {co}")
        want.append(f"linear {lib.s3r_linear_scratch_elems(b, ci, co)}")
    out = subprocess.run([consumer], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    got = out.strip().split("\n")[1:]
    assert got == want
    assert sum(1 for g in got if g.startswith("desc")) == 8 + 10 + 3
    assert all(int(g.split()[1]) > 0 for g in got if g.startswith("chain"))


def test_c_consumer_gets_errors_not_crashes(consumer):
    """a nonsense descriptor through the C side: negative sizes / codes and a message, never a crash"""
    bad = "desc 0 2 2 32 64 8 3 0 1 1 0 -1 1 1 0 0 0 0 0 1 0 0.0\n"            # stride 0
    r = subprocess.run([consumer], input=bad + "chain 1\n" + bad, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")[1:]
    assert lines[0].split()[1] == "-1" and lines[1].startswith("chain -1 ")


def _integration_stub():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "s3r_abi_version" in b]
    assert len(stub) == 1
    return stub[0]


def test_integration_md_stub_runs(s3r, lib, monkeypatch):
    """INTEGRATION.md §B's ctypes stub, up to its 'a GPU is needed' line, executes against the built library: the version it
    asserts, the struct it declares and the planning calls it makes are the library's (the ABI-7 stub had rotted)"""
    stub = _integration_stub()
    head, marker, tail = stub.partition("# --- from here on a GPU is needed")
    assert marker and "s3r_encoder_forward" in tail
    monkeypatch.chdir(ROOT)
    ns = {}
    exec(compile(head, "INTEGRATION.md", "exec"), ns)
    L = s3r._lib
    assert [n for n, _ in ns["ConvDesc"]._fields_] == [n for n, _ in L.ConvDesc._fields_]
    assert [(n, t) for n, t in ns["ConvDesc"]._fields_] == [(n, t) for n, t in L.ConvDesc._fields_]
    assert C.sizeof(ns["ConvDesc"]) == C.sizeof(L.ConvDesc) == 88 and C.sizeof(ns["Layer"]) == C.sizeof(L.Layer) == 112
    assert ns["scratch_elems"] > 0 and ns["packed_elems"].value > 0
    # every entry point the stub names exists, and the argument counts of the calls it shows are the header's
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "s3r.h")).read(), flags=re.S)
    import ast
    calls = [n for n in ast.walk(ast.parse(stub)) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
             and isinstance(n.func.value, ast.Name) and n.func.value.id == "lib"]
    assert len(calls) >= 8
    for call in calls:
        name = call.func.attr
        proto = re.search(r"\b%s\s*\(([^)]*)\)" % name, header)
        assert proto, name
        n_proto = 0 if proto.group(1).strip() == "void" else proto.group(1).count(",") + 1
        assert len(call.args) == n_proto, (name, len(call.args), n_proto)


def test_no_stale_abi_numbers_in_the_docs():
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert not re.search(r"s3r_abi_version\(\)\s*==\s*[0-7]\b", text), doc
