"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding / timing / gather helpers that
bench.py and multi-GPU callers use (no GPU, no compute calls)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import importlib
    par = importlib.import_module("zkt_plonk_amd.parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = par.shard_range(7, rank, world)
        proofs = [bytes([u]) * 802 for u in range(lo, hi)]      # stand-ins for 802-byte proofs
        par.barrier(dist)
        t = par.max_over_ranks(dist, 1.0 + rank)
        allp = par.gather_proofs(dist, proofs)
        q.put((rank, lo, hi, t, [p[0] for p in allp], all(len(p) == 802 for p in allp)))
    finally:
        dist.destroy_process_group()


def test_world_size_two_sharding_and_timing():
    sys.path.insert(0, ROOT)
    import zkt_plonk_amd  # noqa: F401  (registers the package so the workers can import the submodule)
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    assert all(abs(r[3] - 2.0) < 1e-9 for r in res)            # max over ranks
    assert all(r[4] == list(range(7)) and r[5] for r in res)   # global unit order on every rank


def test_shard_range_covers_everything():
    import importlib
    sys.path.insert(0, ROOT)
    import zkt_plonk_amd  # noqa: F401
    par = importlib.import_module("zkt_plonk_amd.parallel")
    for total in (0, 1, 5, 8, 13):
        for world in (1, 2, 3, 8):
            spans = [par.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
