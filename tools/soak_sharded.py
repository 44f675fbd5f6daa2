"""Soak run of ONE proof sharded over `world` contexts on one GPU (threads + in-process all-gather): every rank's
proof must equal the single-context bytes, over hundreds of chained proofs.  usage: soak_sharded.py [world] [log_n] [proofs]"""
import sys, os, random, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench as B
import zkt_plonk_amd as z
from zkt_plonk_amd import parallel as par

world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
total = int(sys.argv[3]) if len(sys.argv) > 3 else 200
fld = B.FIELDS["bn254"]
n = 1 << log_n
L = fld["fq_limbs"]
tau = 424243
ref = z.Context("bn254", 0)
ref.srs_generate(tau, n + 8)
circs = [B.synthetic_circuit(fld, log_n, table_size=64, value_seed=20 + k) for k in range(2)]
evals = {name: B.fr_to_mont_gpu(ref, fld, circs[0]["sel"][name]) for name in z.PK_ORDER}
prover, commits = z.GpuProver.setup(ref, log_n, evals)
rinv_q = pow(1 << (64 * L), -1, fld["q"])
vk = {}
for name in z.PK_ORDER:
    xy, inf = commits[name]
    vk[name] = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv_q % fld["q"],
                                 sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv_q % fld["q"])
gates = circs[0]["gates"]
rnd = random.Random(8)
jobs = []
for circ in circs:
    hw = [B.fr_to_mont_gpu(ref, fld, circ[w][:gates]) for w in "abc"]
    pi_pos = sorted(circ["pi"])
    jobs.append((hw[0], hw[1], hw[2], B.fr_to_mont_gpu(ref, fld, circ["table"]), pi_pos,
                 B.fr_to_mont_gpu(ref, fld, [circ["pi"][i] for i in pi_pos]),
                 B.fr_to_mont_gpu(ref, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])))

def tr():
    t = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
    return z.seed_transcript(t, n, vk)

want = [ref.prove(*job, tr()) for job in jobs]
srs = ref.srs_download(0, n + 8)
group = par.LocalGroup(world)
bad = [0] * world
order = [rnd.randrange(2) for _ in range(total + 1)]

def rank_main(rank):
    try:
        ctx = z.Context("bn254", 0)
        ctx.set_comm(group.comm(rank))
        lo, hi = par.shard_range(n + 8, rank, world)
        ctx.srs_load_slice(srs[lo:hi], lo, n + 8)
        z.GpuProver.setup(ctx, log_n, evals)
        preps = [ctx.prepare_host(*job) for job in jobs]
        for i in range(total):
            got = ctx.prove_prepared(preps[order[i]], tr(), preps[order[i + 1]])
            if got != want[order[i]]:
                bad[rank] += 1
        ctx.close()
    except BaseException:
        bad[rank] += 1000000
        group.barrier.abort()
        raise

t0 = time.time()
ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in ths: t.start()
for t in ths: t.join()
print("SHARDED SOAK %s: world %d, n=2^%d, %d proofs per rank, mismatches %s, %.1f s" % (
    "OK" if not any(bad) else "FAILED", world, log_n, total, bad, time.time() - t0), flush=True)
sys.exit(1 if any(bad) else 0)
