"""The N>1 path on CPU: batch sharding + the single eval-collation all-gather, world_size 2 over gloo
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


@pytest.mark.parametrize("total", [8, 7, 1, 0])
def test_sharded_forward_world2_gloo(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
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
