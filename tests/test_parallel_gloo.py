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


def _msm_worker(rank, world, port, q):
    """Index-range-sharded MSM: each rank's partial sum comes from the CPU oracle here (the test has no GPU); the
    exchange and the host-side point addition are the product's."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    import importlib
    par = importlib.import_module("zkt_plonk_amd.parallel")
    from oracle import coracle as K, fields as F
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cv = F.BN254
        n = 1000
        srs = K.srs_mont(cv, 0xFEED, n)
        rng = np.random.default_rng(11)
        sc = rng.integers(0, 1 << 61, size=(n, 4), dtype=np.uint64)
        sc[:, 3] &= (1 << 60) - 1
        lo, hi = par.shard_range(n, rank, world)
        part, pinf = K.msm_mont(cv, srs[lo:hi], sc[lo:hi], True)          # this rank's slice of points and scalars
        total, tinf = par.sharded_msm_combine(dist, "bn254", part, pinf)
        want, winf = K.msm_mont(cv, srs, sc, True)
        # an all-identity contribution must be neutral
        zero, zinf = par.sharded_msm_combine(dist, "bn254", part, True)
        q.put((rank, bool(tinf == winf and np.array_equal(total, want)), bool(zinf)))
    finally:
        dist.destroy_process_group()


def test_world_size_two_sharded_msm_combine():
    sys.path.insert(0, ROOT)
    import zkt_plonk_amd  # noqa: F401
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_msm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)


def test_g1_sum_host_matches_oracle():
    sys.path.insert(0, ROOT)
    import numpy as np
    import zkt_plonk_amd as z
    from zkt_plonk_amd._lib import g1_sum_host
    from oracle import coracle as K, fields as F, curve as C
    for cv in (F.BN254, F.BLS12_381):
        srs = K.srs_mont(cv, 0xBEEF, 9)
        pts = K.points_from_mont(cv, srs)
        acc = None
        for P_ in pts:
            acc = C.add(cv, acc, P_)
        out, inf = g1_sum_host(cv.name, srs)
        assert not inf and K.points_from_mont(cv, out.reshape(1, -1))[0] == acc
        # P + (-P) and the empty sum are the identity; a doubled point takes the doubling path
        neg = srs[:1].copy()
        L = cv.fq.limbs64
        y = sum(int(v) << (64 * i) for i, v in enumerate(neg[0, L:]))
        ny = (cv.fq.p - y) % cv.fq.p
        neg[0, L:] = [(ny >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(L)]
        assert g1_sum_host(cv.name, np.vstack([srs[:1], neg]))[1]
        assert g1_sum_host(cv.name, np.zeros((0, 2 * L), dtype=np.uint64))[1]
        out, inf = g1_sum_host(cv.name, np.vstack([srs[1:2], srs[1:2]]))
        assert K.points_from_mont(cv, out.reshape(1, -1))[0] == C.add(cv, pts[1], pts[1])


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


def _comm_worker(rank, world, port, q):
    """The communicator a sharded proof uses (parallel.TorchComm -> zkt_comm_vtable), driven from the C side through
    zkt_comm_selftest: the bytes of every rank come back in rank order on every rank."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import importlib
    par = importlib.import_module("zkt_plonk_amd.parallel")
    lib = importlib.import_module("zkt_plonk_amd._lib")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = par.TorchComm(dist)
        ok = True
        for nbytes in (1, 128, 6 * 192, 70000):
            mine = bytes((rank * 37 + i) & 0xFF for i in range(nbytes))
            got = lib.comm_selftest(comm.vt, mine)
            want = b"".join(bytes((r * 37 + i) & 0xFF for i in range(nbytes)) for r in range(world))
            ok = ok and got == want
        q.put((rank, ok, comm.vt.world, comm.vt.device_buffers, comm.calls))
    finally:
        dist.destroy_process_group()


def test_world_size_two_comm_vtable_round_trip():
    sys.path.insert(0, ROOT)
    import zkt_plonk_amd  # noqa: F401
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] == 2 and r[3] == 0 and r[4] == 4 for r in res)


def test_class_decomposition_of_the_coset_matches_the_oracle():
    """The index algebra a sharded proof rests on, with the oracle's transforms (no GPU): the outputs of the 4n coset
    transform whose index is cls mod G are the coset transform of size 4n / G with shift g w_4n^cls of the polynomial
    folded modulo X^(4n/G) - shift^(4n/G); "omega-next" (index + 4) stays in the class for G <= 4 and moves to class
    (cls + 4) mod 8, entry i + (cls + 4) // 8, for G = 8."""
    sys.path.insert(0, ROOT)
    from oracle import fields as F
    from oracle.ntt import Domain
    from helpers import field_elems
    cv = F.BN254
    p = cv.fr.p
    n = 16
    big = Domain(cv.fr, 4 * n)
    coeffs = field_elems(p, 5, n + 8)
    full = big.coset_fft(coeffs)
    w4n = big.group_gen
    for G in (1, 2, 4, 8):
        m = 4 * n // G
        sub = Domain(cv.fr, m)
        for cls in range(G):
            shift = cv.fr.generator * pow(w4n, cls, p) % p
            cm = pow(shift, m, p)
            folded = [0] * m
            for i, v in enumerate(coeffs):
                folded[i % m] = (folded[i % m] + v * pow(cm, i // m, p)) % p
            scaled = [v * pow(shift, i, p) % p for i, v in enumerate(folded)]
            got = sub.fft(scaled)
            assert got == full[cls::G], (G, cls)
            # omega-next
            for i in range(m):
                t = (cls + G * i + 4) % (4 * n)
                if G <= 4:
                    assert t % G == cls and t // G == (i + 4 // G) % m
                else:
                    assert t % G == (cls + 4) % 8 and t // G == (i + (cls + 4) // 8) % m
