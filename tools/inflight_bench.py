"""Experiment: throughput with F proofs in flight on one GPU (F independent contexts driven by F host threads).
usage: python tools/inflight_bench.py F steps [shared|own] [log_n]"""
import sys, os, time, threading, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench as B
import zkt_plonk_amd as z

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shared = len(sys.argv) > 3 and sys.argv[3] == "shared"   # all contexts enqueue on ONE stream: kernels never overlap
shared_stream = None
log_n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
fld = B.FIELDS["bn254"]
n = 1 << log_n
dev = torch.device("cuda", 0)
circ = B.synthetic_circuit(fld, log_n)
L = fld["fq_limbs"]
workers = []
for w in range(F):
    ctx = z.Context("bn254", 0)
    if shared:
        shared_stream = shared_stream or torch.cuda.Stream(dev)
        s = shared_stream
    else:
        s = torch.cuda.Stream(dev)
    ctx.set_stream(s.cuda_stream)
    ctx.srs_generate(0x5EED5EED1234567890ABCDEF % fld["r"], n + 8)
    pk = {name: ctx.ntt(log_n, B.fr_to_mont_gpu(ctx, fld, circ["sel"][name]), inverse=True) for name in z.PK_ORDER}
    rinv_q = pow(1 << (64 * L), -1, fld["q"])
    vk = {}
    for name in z.PK_ORDER:
        xy, inf = ctx.msm(pk[name])
        vk[name] = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv_q % fld["q"],
                                     sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv_q % fld["q"])
    prover = z.GpuProver(ctx, log_n, pk)
    gates = circ["gates"]
    wires = [torch.from_numpy(B.fr_to_mont_gpu(ctx, fld, circ[k][:gates]).view(np.int64)).to(dev) for k in "abc"]
    table = B.fr_to_mont_gpu(ctx, fld, circ["table"])
    pi_pos = sorted(circ["pi"])
    pi_vals = B.fr_to_mont_gpu(ctx, fld, [circ["pi"][k] for k in pi_pos])
    rnd = random.Random(99)
    blinders = B.fr_to_mont_gpu(ctx, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])
    prep = ctx.prepare_dev(wires[0].data_ptr(), wires[1].data_ptr(), wires[2].data_ptr(), gates, table, pi_pos, pi_vals, blinders)
    def one(ctx=ctx, prep=prep, vk=vk, _keep=wires):   # the prepared inputs point into these tensors
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
        z.seed_transcript(tr, n, vk)
        return ctx.prove_prepared(prep, tr, prep)      # chained: every proof announces the next one
    workers.append((ctx, prover, one))
proofs = [None] * F
def run(w, k):
    for _ in range(k):
        proofs[w] = workers[w][2]()
for w in range(F):
    run(w, 2)
torch.cuda.synchronize()
t = time.perf_counter()
th = [threading.Thread(target=run, args=(w, steps // F)) for w in range(F)]
for x in th: x.start()
for x in th: x.join()
torch.cuda.synchronize()
el = time.perf_counter() - t
total = (steps // F) * F
print(("shared-stream " if shared else "") + "n=2^%d inflight=%d: %d proofs in %.3f s -> %.2f proofs/s (%.2f ms/proof); proofs identical: %s" % (
    log_n, F, total, el, total / el, 1e3 * el / total, all(p == proofs[0] for p in proofs)), flush=True)
