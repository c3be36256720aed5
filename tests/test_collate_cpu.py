"""The N>1 path on CPU: batch sharding + the single eval-collation all-gather over gloo, at world_size 2 and at the real
world size of BASELINE configs[4] (8 ranks, ragged shards: 7 and 9 samples leave ranks empty / uneven, 2047 = 8 x 256 - 1)
(the same code runs over RCCL/xGMI with backend "nccl" on the GPU box: bench.py --gpus N)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_forward(left, right):
    """Stands in for the HIP forward (no GPU here): a per-sample function of both inputs, so any
    mis-ordered, dropped or duplicated sample changes the collated result."""
    return (left.flatten(1).sum(1, keepdim=True) * 3 + right.flatten(1).sum(1, keepdim=True)).repeat(1, 5)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import s3r
    torch.set_num_threads(1)                 # (8 ranks on the container's 8 CPUs: one host thread each)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        left, right = torch.rand(total, 3, 4, 4, generator=g), torch.rand(total, 3, 4, 4, generator=g)
        got = s3r.collate.sharded_forward(_fake_forward, left, right)
        want = _fake_forward(left, right)
        b, e = s3r.collate.shard_bounds(total, world, rank)
        local = s3r.collate.sharded_forward(_fake_forward, left, right, gather=False)
        ok = torch.equal(got, want) and torch.equal(local, want[b:e])
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok), tuple(got.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (2, 7), (2, 1), (2, 0), (8, 7), (8, 9), (8, 2047)])
def test_sharded_forward_gloo(world, total):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True] * world, res
    assert all(r[2] == (total, 5) for r in res)


def test_shard_bounds_cover_every_sample_once(s3r):
    for total in (0, 1, 5, 32, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [s3r.collate.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == s3r.collate.shard_sizes(total, world)
    with pytest.raises(ValueError):
        s3r.collate.shard_bounds(4, 2, 2)


def test_bench_and_runner_start_their_own_ranks(monkeypatch):
    """`python bench.py --gpus N` / `python runner.py --test --gpus N` with no launcher around them (how the driver
    invokes N = 1, and what it would do for N > 1): the parent starts `torch.distributed.run` with N ranks as a CHILD
    process — before importing torch — on 127.0.0.1, passes its own arguments through, and returns the child's code."""
    import importlib
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    for name, argv in (("bench", ["bench.py", "--gpus", "4", "--steps", "7", "--batch", "256"]),
                       ("runner", ["runner.py", "--test", "--gpus", "4", "--samples", "8"])):
        mod = importlib.import_module(name)
        calls = []

        class Done:
            returncode = 17

        def fake_run(cmd, env=None, **kw):
            calls.append((cmd, env))
            return Done()
        monkeypatch.setattr(mod.subprocess, "run", fake_run)
        monkeypatch.setattr(sys, "argv", argv)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            monkeypatch.delenv(k, raising=False)
        with pytest.raises(SystemExit) as e:
            mod.main()
        assert e.value.code == 17                                   # the child's exit code, relayed
        (cmd, env), = calls
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-len(argv) + 1:] == argv[1:] and cmd[-len(argv)].endswith(argv[0])
        assert env["MASTER_ADDR"] == "127.0.0.1"
        # under an external launcher (WORLD_SIZE set) nothing is spawned: the script is one of the ranks.  The first thing
        # a rank does is pick its device: stop it THERE with a sentinel (nothing of the GPU or of torch.distributed is
        # initialised in this process, with or without a GPU / MASTER_PORT in the environment)
        import torch

        class _Reached(Exception):
            pass

        def stop(*a, **k):
            raise _Reached()
        monkeypatch.setattr(torch.cuda, "set_device", stop)
        monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
        monkeypatch.setattr(torch.distributed, "init_process_group", stop)
        monkeypatch.setenv("WORLD_SIZE", "4")
        monkeypatch.delenv("MASTER_ADDR", raising=False)
        monkeypatch.delenv("MASTER_PORT", raising=False)
        calls.clear()
        with pytest.raises(_Reached):
            mod.main()
        assert not calls
        monkeypatch.delenv("WORLD_SIZE")


def test_bench_override_flag_parses_into_the_debug_context(s3r, monkeypatch):
    """`bench.py --override tile.v2=2 ksplit.v4=2 algo.e6=1` -> s3r.debug_overrides kwargs; a run under it (or under a kernel A/B
    environment switch) is listed in the line's `kernel_env_overrides`, i.e. marked as not the plain configuration."""
    import bench
    kw = bench.parse_overrides(["tile.v2=2", "ksplit.v4=2", "algo.e6=1"])
    assert kw == {"tile": {"v2": 2}, "ksplit": {"v4": 2}, "algo": {"e6": 1}}
    with pytest.raises(SystemExit):
        bench.parse_overrides(["tiles.v2=2"])
    with pytest.raises(SystemExit):
        bench.parse_overrides(["tile.v2=two"])
    for k in list(os.environ):
        if k.startswith("S3R_"):
            monkeypatch.delenv(k)
    assert bench.kernel_env_overrides(s3r) == []
    monkeypatch.setenv("S3R_WINO", "0")
    with s3r.debug_overrides(**kw):
        assert bench.kernel_env_overrides(s3r) == ["S3R_WINO", "algo.e6=1", "ksplit.v4=2", "tile.v2=2"]
    assert bench.kernel_env_overrides(s3r) == ["S3R_WINO"]
