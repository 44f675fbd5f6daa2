"""Debug aid: several contexts alive in one process, proofs checked against the first context's bytes."""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench as B
import zkt_plonk_amd as z

F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
fld = B.FIELDS["bn254"]
n = 1 << log_n
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
circ = B.synthetic_circuit(fld, log_n)
L = fld["fq_limbs"]
ref = None
keep = []
for w in range(F):
    ctx = z.Context("bn254", 0)
    ctx.srs_generate(0x5EED5EED1234567890ABCDEF % fld["r"], n + 8)
    evals = {name: B.fr_to_mont_gpu(ctx, fld, circ["sel"][name]) for name in z.PK_ORDER}
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    rinv_q = pow(1 << (64 * L), -1, fld["q"])
    vk = {}
    for name in z.PK_ORDER:
        xy, inf = commits[name]
        vk[name] = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv_q % fld["q"],
                                     sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv_q % fld["q"])
    gates = circ["gates"]
    hw = [B.fr_to_mont_gpu(ctx, fld, circ[k][:gates]) for k in "abc"]
    table = B.fr_to_mont_gpu(ctx, fld, circ["table"])
    pi_pos = sorted(circ["pi"])
    pi_vals = B.fr_to_mont_gpu(ctx, fld, [circ["pi"][k] for k in pi_pos])
    rnd = random.Random(99)
    blinders = B.fr_to_mont_gpu(ctx, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])
    for rep in range(2):
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
        z.seed_transcript(tr, n, vk)
        try:
            pr = ctx.prove(hw[0], hw[1], hw[2], table, pi_pos, pi_vals, blinders, tr)
        except Exception as e:
            print("ctx %d rep %d: ERROR %s" % (w, rep, e), flush=True)
            continue
        if ref is None:
            ref = pr
        print("ctx %d rep %d: %s  vk_same=%s" % (w, rep, "same" if pr == ref else "DIFFERENT", True), flush=True)
    if mode == "garbage":   # leave dirty memory behind for the next context
        junk = torch.full((1 << 28,), -1, dtype=torch.int64, device=dev)
        del junk
        torch.cuda.empty_cache()
    keep.append((ctx, prover))
