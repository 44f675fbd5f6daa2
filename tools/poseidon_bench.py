"""Throughput of the device Poseidon permutation (zkt_poseidon_hash_batch_dev) on the reference's BN254 parameter sets
(tests/golden/poseidon_bn254.npz): hashes/s with and without the per-round states.  usage (GPU box):
python tools/poseidon_bench.py > gpurun_out/poseidon.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkt_plonk_amd as z
import bench as B

arr = np.load(os.path.join(ROOT, "tests", "golden", "poseidon_bn254.npz"))
fld = B.FIELDS["bn254"]
ctx = z.Context("bn254", 0)
mont = lambda canon: B.fr_to_mont_gpu(ctx, fld, [sum(int(v) << (64 * i) for i, v in enumerate(r)) for r in canon])
batch = 1 << 18
print("device Poseidon, BN254, reference parameter sets, batch = %d hashes per launch (one thread per hash)" % batch)
for w, partial in ((3, 55), (4, 56), (5, 56)):
    rounds = 8 + partial
    h = ctx.poseidon_load(w, 4, partial, mont(arr["rc_x%d" % w]), mont(arr["mds_x%d" % w]), mont([[(1 << (w - 1)) - 1, 0, 0, 0]])[0])
    rng = np.random.default_rng(w)
    ins = rng.integers(0, 1 << 62, size=(batch * (w - 1), 4), dtype=np.uint64)
    ins[:, 3] >>= np.uint64(2)
    d_in, d_out, d_st = ctx.alloc(ins.nbytes), ctx.alloc(batch * 32), ctx.alloc(batch * (rounds + 1) * w * 32)
    ctx.upload(d_in, ins)
    for states in (0, d_st):
        ctx.poseidon_hash_batch_dev(h, d_in, batch, w - 1, d_out, states)
        ctx.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.poseidon_hash_batch_dev(h, d_in, batch, w - 1, d_out, states)
        ctx.synchronize()
        dt = (time.perf_counter() - t) / reps
        muls = 8 * (3 * w + w * w) + partial * (3 + w * w)       # field products per hash (SURVEY.md 8d.4 gate formula)
        print("x%d (Rf 8, Rp %d) %-14s %8.3f ms  %10.3e hashes/s  %9.3e field products/s%s" % (
            w, partial, "with states" if states else "hashes only", dt * 1e3, batch / dt, batch * muls / dt,
            "  (%.1f GB of states written)" % (batch * (rounds + 1) * w * 32 / 1e9) if states else ""))
    # the gadget's witness (zkt_poseidon_gadget_witness_dev): vars_per_hash variables per hash, 32 B each
    per = ctx.poseidon_gadget_vars_per_hash(h)
    for kname, kernel, gb in (("one thread per hash", 1, 1 << 18), ("one thread per hash", 1, 538), ("W^2 lanes per hash", 2, 538),
                              ("W^2 lanes per hash", 2, 1 << 14)):
        d_vars = ctx.alloc(gb * per * 32)
        run = lambda: ctx.poseidon_gadget_witness_dev(h, gb, w - 1, d_vars, gb * per, d_inputs=d_in, kernel=kernel)
        run()
        ctx.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            run()
        ctx.synchronize()
        dt = (time.perf_counter() - t) / reps
        print("x%d gadget witness, %-20s batch %6d: %8.3f ms  %10.3e hashes/s  %9.3e products/s  %6.1f GB/s of variables written" % (
            w, kname, gb, dt * 1e3, gb / dt, gb * per / dt, gb * per * 32 / dt / 1e9))
        ctx.free(d_vars)
    ctx.poseidon_free(h)
    for d in (d_in, d_out, d_st):
        ctx.free(d)
ctx.close()
