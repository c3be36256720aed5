"""The GPU leg of the compiled-C consumer (tests/c_abi/consumer_gpu.c): one s3r_conv_forward, one ConvTranspose3d and one
s3r_chamfer_forward driven from plain C through include/s3r.h — device memory from a dlopen'ed libamdhip64, no torch, no ctypes in
the measured process — compared with the committed vectors under tests/golden/ (oracle-made: parity unpinned)."""
import os
import subprocess

import numpy as np
import pytest

from tests.test_c_abi_cpu import build_consumer, desc_line

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def consumer_gpu(s3r):
    s3r.load_library()
    return build_consumer("consumer_gpu", extra=("-ldl",))


def run(exe, *args):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    return r.stdout


@pytest.mark.parametrize("case,algo", [("c2d", "direct"), ("c2d", "winograd"), ("c2d", "auto"), ("dc3", "direct"), ("dc3", "auto")])
def test_conv_forward_from_c_matches_golden(s3r, consumer_gpu, golden_dir, tmp_path, case, algo):
    from tests.golden.make_conv_kat import CASES
    layer, batch, n = CASES[case]
    z = np.load(os.path.join(golden_dir, "conv_kat.npz"))
    nd = s3r.arch_spec.ndim(layer)
    x = np.pad(z[f"{case}_x"], [(0, 0), (0, 0)] + [(1, 1)] * nd)              # halo 1: the MFMA kernels read their padding from memory
    scale = z[f"{case}_gamma"] / np.sqrt(z[f"{case}_var"] + np.float32(s3r.arch_spec.BN_EPS))
    shift = z[f"{case}_beta"] + (z[f"{case}_b"] - z[f"{case}_mean"]) * scale
    for name, a in (("x", x), ("w", z[f"{case}_w"]), ("scale", scale), ("shift", shift)):
        np.ascontiguousarray(a, dtype=np.float32).tofile(tmp_path / f"{name}.bin")
    d = s3r._lib.make_desc(layer, batch, n, in_halo=1, out_halo=0, algo=s3r._lib.ALGO[algo])
    out = run(consumer_gpu, "conv", tmp_path, *desc_line(d).split()[1:])
    want = z[f"{case}_y"]
    assert f"y_elems={want.size}" in out
    got = np.fromfile(tmp_path / "y.bin", dtype=np.float32).reshape(want.shape)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 1e-5 and np.abs(got - want).max() < 1e-4, (rel, np.abs(got - want).max())      # fp32 bar: north_star's 1e-4 relative


def test_chamfer_forward_from_c_matches_golden(consumer_gpu, golden_dir, tmp_path):
    import torch
    z = np.load(os.path.join(golden_dir, "s2p_chamfer.npz"))
    # the known-answer clouds, as stored
    z["kat_p"].tofile(tmp_path / "p.bin")
    z["kat_q"].tofile(tmp_path / "q.bin")
    run(consumer_gpu, "chamfer", tmp_path, 1, 3, 2)
    for name, key, dt in (("d1", "kat_d1", np.float32), ("d2", "kat_d2", np.float32), ("i1", "kat_i1", np.int32), ("i2", "kat_i2", np.int32)):
        assert np.array_equal(np.fromfile(tmp_path / f"{name}.bin", dtype=dt).reshape(z[key].shape), z[key]), name
    # the seeded pair of tests/golden/make_golden.py (generator seed 4: p then q)
    g = torch.Generator().manual_seed(4)
    p, q = torch.rand(2, 256, 3, generator=g), torch.rand(2, 300, 3, generator=g)
    p.numpy().tofile(tmp_path / "p.bin")
    q.numpy().tofile(tmp_path / "q.bin")
    run(consumer_gpu, "chamfer", tmp_path, 2, 256, 300)
    for name, key, dt in (("d1", "rnd_d1", np.float32), ("d2", "rnd_d2", np.float32), ("i1", "rnd_i1", np.int32), ("i2", "rnd_i2", np.int32)):
        assert np.array_equal(np.fromfile(tmp_path / f"{name}.bin", dtype=dt).reshape(z[key].shape), z[key]), name
