"""Shared helpers for the test-suite: deterministic inputs (splitmix64) and golden decoding."""
import hashlib

M64 = (1 << 64) - 1


def splitmix64(seed):
    s = seed & M64
    while True:
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def field_elems(p, seed, count):
    g = splitmix64(seed)
    out = []
    for _ in range(count):
        v = 0
        for i in range(4):
            v |= next(g) << (64 * i)
        out.append(v % p)
    return out


def digest(vals):
    return hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in vals)).hexdigest()


def unhex_point(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def rand_fr(rng, n):
    """n valid field elements as (n, 4) uint64 Montgomery words: any value below 2^253 is below both
    scalar moduli, so random limbs with the top limb < 2^61 are canonical representatives."""
    import numpy as np
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    return x


def sharded_rank_job(z, par, comm, cv, n, srs_arr, evals, vk, jobs, out, rank, world, use_setup):
    """What one rank of a sharded prover does: its SRS slice, the circuit (laid out for its class of the
    4n coset), the proofs.  out[rank] = (proofs, comm_stats) or the exception."""
    try:
        ctx = z.Context(cv.name, 0)
        ctx.set_comm(comm)
        lo, hi = par.shard_range(n + 8, rank, world)
        ctx.srs_load_slice(srs_arr[lo:hi], lo, n + 8)
        log_n = n.bit_length() - 1
        if use_setup:
            prover, commits = z.GpuProver.setup(ctx, log_n, evals["evals"])
            L = cv.fq.limbs64
            rinv = pow(1 << (64 * L), -1, cv.fq.p)
            for name in z.PK_ORDER:
                xy, inf = commits[name]
                pt = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv % cv.fq.p,
                                       sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv % cv.fq.p)
                assert pt == vk.commits[name], name
        else:
            z.GpuProver(ctx, log_n, evals["pk"])
        setup_stats = ctx.comm_stats()
        proofs = []
        preps = [ctx.prepare_host(*job) for job in jobs]
        for i, prep in enumerate(preps):
            tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
            z.seed_transcript(tr, vk.n, vk.commits)
            proofs.append(ctx.prove_prepared(prep, tr, preps[i + 1] if i + 1 < len(preps) else None))
        out[rank] = (proofs, ctx.comm_stats(), setup_stats)
        ctx.close()
    except BaseException as e:          # a rank that dies must not leave the others waiting at the barrier forever
        out[rank] = e
        try:
            comm.group.barrier.abort()
        except Exception:
            pass
        raise


def run_sharded_ranks(z, cv, n, srs_arr, evals, vk, jobs, world, use_setup=True, timeout=900):
    """`world` contexts on ONE GPU play the ranks of a sharded proof (threads + the in-process all-gather of
    parallel.LocalGroup).  Returns [(proofs, (calls, bytes_sent), (setup_calls, setup_bytes))] per rank."""
    import threading
    from zkt_plonk_amd import parallel as par
    group = par.LocalGroup(world)
    out = [None] * world
    ths = [threading.Thread(target=sharded_rank_job, args=(z, par, group.comm(r), cv, n, srs_arr, evals, vk, jobs, out, r,
                                                          world, use_setup)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=timeout)
    for r in range(world):
        assert not isinstance(out[r], BaseException) and out[r] is not None, out[r]
    return out


def sharded_exchange_bytes(cv, n, world, n_proofs, same_table_proofs=0):
    """Bytes ONE rank sends per the design: per proof four all-gathers of the round's partial sums
    (k = 6, 2, 3, 2 XYZZ points of 4 Fq each; the message has k entries whether or not the table commitment is cached)
    and ONE quotient exchange of exactly 4n * 32 / world bytes."""
    xyzz = 4 * cv.fq.limbs64 * 8
    return n_proofs * (13 * xyzz + (4 * n // world) * 32)
