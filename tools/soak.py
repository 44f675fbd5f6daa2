"""Soak run (development aid): thousands of chained proofs over alternating witnesses and tables, every proof compared
with the bytes the same inputs gave unchained at the start.  A race in the early-round / side-stream logic would show
up as a mismatch.  usage: soak.py [log_n] [proofs]"""
import sys, os, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench as B
import zkt_plonk_amd as z

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fld = B.FIELDS["bn254"]
n = 1 << log_n
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = z.Context("bn254", 0)
ctx.srs_generate(12345, n + 8)
circs = [B.synthetic_circuit(fld, log_n, value_seed=10 + k) for k in range(3)]
evals = {name: B.fr_to_mont_gpu(ctx, fld, circs[0]["sel"][name]) for name in z.PK_ORDER}
prover, commits = z.GpuProver.setup(ctx, log_n, evals)
L = fld["fq_limbs"]
rinv_q = pow(1 << (64 * L), -1, fld["q"])
vk = {}
for name in z.PK_ORDER:
    xy, inf = commits[name]
    vk[name] = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv_q % fld["q"],
                                 sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv_q % fld["q"])
gates = circs[0]["gates"]
rnd = random.Random(5)
preps, keep = [], []
tables = [B.fr_to_mont_gpu(ctx, fld, circs[0]["table"])]
t2 = tables[0].copy(); t2[[0, 1]] = t2[[1, 0]]
tables.append(t2)                       # same set, other order: a different table polynomial
for k, circ in enumerate(circs):
    hw = [B.fr_to_mont_gpu(ctx, fld, circ[w][:gates]) for w in "abc"]
    pi_pos = sorted(circ["pi"])
    pi_vals = B.fr_to_mont_gpu(ctx, fld, [circ["pi"][i] for i in pi_pos])
    bl = B.fr_to_mont_gpu(ctx, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])
    for tb in tables:
        if k == 2:      # the composer's own layout in HBM (variables + wire indices; prove.rs:49-55 runs on the device)
            dv = torch.from_numpy(np.concatenate(hw).view(np.int64)).to(dev)
            di = [torch.from_numpy((np.arange(gates, dtype=np.uint32) + np.uint32(j * gates)).view(np.int32)).to(dev) for j in range(3)]
            keep.append((dv, di))
            preps.append(ctx.prepare_vars_dev(dv.data_ptr(), 3 * gates, di[0].data_ptr(), di[1].data_ptr(), di[2].data_ptr(), gates, tb,
                                              pi_pos, pi_vals, bl))
        elif k % 2 == 0:
            dw = [torch.from_numpy(x.view(np.int64)).to(dev) for x in hw]
            keep.append(dw)
            preps.append(ctx.prepare_dev(dw[0].data_ptr(), dw[1].data_ptr(), dw[2].data_ptr(), gates, tb, pi_pos, pi_vals, bl))
        else:
            preps.append(ctx.prepare_host(hw[0], hw[1], hw[2], tb, pi_pos, pi_vals, bl))

def tr():
    t = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
    return z.seed_transcript(t, n, vk)

want = [ctx.prove_prepared(p, tr()) for p in preps]
assert len(set(want)) == len(want)
t0 = time.time()
bad = 0
order = [rnd.randrange(len(preps)) for _ in range(total + 1)]
for i in range(total):
    k, nxt = order[i], order[i + 1]
    announce = preps[nxt] if rnd.random() < 0.8 else (preps[rnd.randrange(len(preps))] if rnd.random() < 0.5 else None)
    got = ctx.prove_prepared(preps[k], tr(), announce)
    if got != want[k]:
        bad += 1
        print("MISMATCH at proof %d (inputs %d)" % (i, k), flush=True)
    if rnd.random() < 0.02:
        ctx.msm(np.ones((3, 4), dtype=np.uint64))      # an unrelated MSM invalidates early work now and then
    if i % 200 == 199:
        print("%d proofs, %d mismatches, %.1f s" % (i + 1, bad, time.time() - t0), flush=True)
print("SOAK %s: %d proofs, %d mismatches" % ("OK" if bad == 0 else "FAILED", total, bad), flush=True)
sys.exit(1 if bad else 0)
