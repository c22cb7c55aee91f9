"""world_size-2 gloo test (CPU) of the multi-GPU path: ensembles are sharded across ranks with no data-path
collective; distributed calls are limited to the barrier and the max/sum reductions bench.py uses."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [65536, 1025, 3]


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
    from dabgpu import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for n_units in CASES:
        first, count = shard.shard_range(n_units, rank, world)
        mine = torch.zeros(n_units, dtype=torch.int32)
        mine[first:first + count] = 1                   # the ensembles this rank would demodulate
        cover = mine.clone()
        dist.all_reduce(cover)                          # test-only collective: shards must tile the set exactly once
        shard.barrier(dist)
        elapsed = 0.010 * (rank + 1)                    # rank 1 is the slow one
        tmax = shard.max_over_ranks(elapsed, dist)
        total = shard.sum_over_ranks(count, dist)
        q.put((n_units, rank, first, count, bool((cover == 1).all()), tmax, total))
    dist.destroy_process_group()


def test_two_rank_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world * len(CASES)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for n_units in CASES:
        (_, r0, f0, c0, ok0, t0, s0), (_, r1, f1, c1, ok1, t1, s1) = [r for r in res if r[0] == n_units]
        assert ok0 and ok1
        assert f0 == 0 and f1 == c0 and c0 + c1 == n_units and abs(c0 - c1) <= 1
        assert t0 == t1 == pytest.approx(0.020)         # value = units / max-over-ranks time
        assert s0 == s1 == n_units


def test_shard_range_properties():
    sys.path.insert(0, os.path.join(ROOT, "dab-radio_amd"))
    from dabgpu import shard
    for n in (0, 1, 7, 8, 65536):
        for w in (1, 2, 4, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f, c), (f2, _) in zip(spans, spans[1:]):
                assert f + c == f2
    with pytest.raises(ValueError):
        shard.shard_range(8, 2, 2)
