#!/usr/bin/env python3
"""bench.py -- proofs/sec of the MI355X prover hot path on the reference's withdraw circuit.

A "step" is one full pass of proof_system::prove (plonk-core/src/proof_system/prove.rs:59-470) over one
batch of synthetic witness rows already resident in HBM: iNTT(n) of every witness polynomial, their
coset NTT(4n), the fused quotient pass, two grand products, the 12-13 KZG commitments (MSMs of ~n points),
the 12 evaluations and the two openings, with Fiat-Shamir (Merlin) on the host.  Default workload
(`--workload withdraw`): BASELINE.json configs[3] -- WithdrawCircuit<Fr, u64, _, Bn254x5, INPUTS = 8, HEIGHT = 64>
(circuits/src/withdraw.rs:57-150: 538 Poseidon gadgets of 1888 gates, Merkle paths, identifier lookups, 64-bit range and
balance rows; 1 019 498 gates -> n = 2^20, TABLE_SIZE = 1024, 12 public inputs), laid out by tools/withdraw_workload.py
for a synthetic wallet.  Its witness is made where the prover reads it: the host computes the ~4 600 variables that are
not outputs of Poseidon gates (note data, paths, selects; the hash VALUES natively, as bin/src/main.rs:248-271 does) and
the device fills the other 1 015 744 (k_poseidon_gadget, `witness` in the JSON line; outside the timed region, like
`circuit.synthesize` is outside proof_system::prove).  `--workload synthetic` is the r01-r03 workload (random mul / add /
linear rows, a lookup every 16th, 7 public inputs).

Headline (`value`): steady state of a proving service with a queue -- two DISTINCT witnesses of the circuit
alternate, each proof announces its successor (zkt_prove_set_next), witnesses resident in HBM, the lookup
table (part of the circuit) unchanged, so its polynomial / commitment / coset are reused.  `config` says so,
and `latency` reports the other end in the same line: a cold single proof (host witness pointers crossing
PCIe, fresh table, no announcement).

`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N worker processes itself through
torch.distributed.run BEFORE anything touches a GPU; under the driver's own torch.distributed.run launch the
workers are already there.  One process per GPU, independent proofs sharded across ranks, no data-path
collective: "scaling": "weak".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the MSM bucket accumulation), timed
live with HIP events on the stream the kernel runs on; `cpu_baseline` times ONE full proof of the same
workload by the CPU oracle (oracle/fastplonk.py + oracle/coracle.cpp: a port, OpenMP over the host cores)
and checks that its bytes equal the GPU proof's.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIELDS = {
    "bn254": dict(
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        two_adicity=28, gen=5, fq_limbs=4, point_bytes=64, lam=254),
    "bls12_381": dict(
        r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
        q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        two_adicity=32, gen=7, fq_limbs=6, point_bytes=96, lam=255),
}
K1, K2 = 7, 13
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAD_CYCLES = 4.4               # best measured v_mad_u64_u32 issue cost, cycles / wave-instr / SIMD (8 waves resident,
                               # profiles/microbench_r03.txt): the ceiling of frac_of_mad_issue_ceiling (r01-r03 and r05 lines)
MAD_CYCLES_RESIDENT = 4.7      # the same at the 2-4 waves per SIMD the accumulation holds (r04's line used this one;
                               # reported beside it as frac_of_mad_issue_ceiling_at_residency)
CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4


def to_limbs(vals, limbs=4):
    nb = 8 * limbs
    buf = b"".join(int(v).to_bytes(nb, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), limbs).copy()


def fr_to_mont_gpu(ctx, fld, vals):
    """canonical ints -> (n, 4) Montgomery words, converted on the GPU (x * R^2 * R^-1)."""
    arr = to_limbs(vals)
    r2 = pow(1 << 256, 2, fld["r"])
    r2arr = np.tile(to_limbs([r2]), (len(vals), 1))
    return ctx.debug_fr_mul(arr, r2arr)


def synthetic_circuit(fld, log_n, table_size=1024, n_public=7, seed=0x5EED, value_seed=None):
    """Withdraw-shaped synthetic trace of exactly 2^log_n rows: mul / add / linear
    gates chained through copy constraints (output of row i-1 = left input of row i), a lookup row every 16,
    n_public public-input rows.  Pure Python integers; no reference or oracle code involved.
    `seed` fixes the circuit (selectors, permutation, lookup table); `value_seed` draws the witness (free wire
    values, looked-up entries, public inputs) from its own generator, so that several witnesses of ONE circuit
    can be produced (None: one generator for both, the round-1 workload)."""
    import random
    rnd = random.Random(seed)
    vr = rnd if value_seed is None else random.Random(value_seed)
    p = fld["r"]
    n = 1 << log_n
    gates = n - 8
    table = []
    seen = set()
    while len(table) < table_size - 1:
        v = rnd.randrange(1, p)
        if v not in seen:
            seen.add(v)
            table.append(v)
    w = pow(fld["gen"], (p - 1) >> log_n, p)
    roots = [1] * n
    for i in range(1, n):
        roots[i] = roots[i - 1] * w % p
    a = [0] * n; b = [0] * n; c = [0] * n
    q_m = [0] * n; q_l = [0] * n; q_r = [0] * n; q_o = [0] * n; q_c = [0] * n; q_lk = [0] * n
    s1 = list(roots)
    s2 = [K1 * x % p for x in roots]
    s3 = [K2 * x % p for x in roots]
    pi = {}
    prev_out = None  # row whose output wire is copied into this row's left wire
    rr = rnd.randrange
    vrr = vr.randrange
    for i in range(gates - n_public):
        if i % 16 == 15:
            t = table[vrr(len(table))]
            a[i] = t; c[i] = t
            q_l[i] = 1; q_o[i] = p - 1; q_lk[i] = 1
        else:
            if prev_out is None:
                a[i] = vrr(p)
            else:
                a[i] = c[prev_out]
                # one permutation cycle {Output(prev_out), Left(i)}
                s3[prev_out] = roots[i]
                s1[i] = K2 * roots[prev_out] % p
            b[i] = vrr(p)
            k = i % 3
            if k == 0:
                c[i] = a[i] * b[i] % p
                q_m[i] = 1; q_o[i] = p - 1
            elif k == 1:
                c[i] = (a[i] + b[i]) % p
                q_l[i] = 1; q_r[i] = 1; q_o[i] = p - 1
            else:
                ql, qr, qc = rr(p), rr(p), rr(p)
                c[i] = (ql * a[i] + qr * b[i] + qc) % p
                q_l[i] = ql; q_r[i] = qr; q_o[i] = p - 1; q_c[i] = qc
        prev_out = i
    for i in range(gates - n_public, gates):
        v = vrr(p)
        c[i] = v
        q_o[i] = p - 1
        pi[i] = v
    q_table = [0] * table_size + [1] * (n - table_size)
    sel = dict(q_m=q_m, q_l=q_l, q_r=q_r, q_o=q_o, q_c=q_c, sigma1=s1, sigma2=s2, sigma3=s3, q_lookup=q_lk,
               q_table=q_table)
    return dict(n=n, gates=gates, a=a, b=b, c=c, sel=sel, table=table, pi=pi)


WITHDRAW_SHAPES = {14: (4, 1, 7), 18: (4, 3, 48), 19: (5, 4, 64), 20: (5, 8, 64), 22: (5, 32, 64)}   # width, INPUTS, HEIGHT


def synthetic_workload(z, torch, ctx, dev, fld, args, log_n):
    """ONE circuit, TWO witnesses (free wire values, looked-up entries and public inputs all differ); wires resident in HBM."""
    import random
    circs = [synthetic_circuit(fld, log_n, value_seed=1 + k) for k in range(2)]
    assert circs[0]["sel"] == circs[1]["sel"] and circs[0]["table"] == circs[1]["table"] and circs[0]["pi"] != circs[1]["pi"]
    # proof_system::setup on the device (setup.rs:42-166): selector / sigma / table-mask evaluations -> ProverKey,
    # ExtendedProverKey and the ten VerifierKey commitments that seed the transcript
    evals = {name: fr_to_mont_gpu(ctx, fld, circs[0]["sel"][name]) for name in z.PK_ORDER}
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    gates = circs[0]["gates"]
    table = fr_to_mont_gpu(ctx, fld, circs[0]["table"])
    rnd = random.Random(99)
    host_w, keep, preps, pis = [], [], [], []
    for circ in circs:
        hw = [fr_to_mont_gpu(ctx, fld, circ[k][:gates]) for k in "abc"]
        dw = [torch.from_numpy(x.view(np.int64)).to(dev) for x in hw]
        pi_pos = sorted(circ["pi"])
        pi_vals = fr_to_mont_gpu(ctx, fld, [circ["pi"][k] for k in pi_pos])
        blinders = fr_to_mont_gpu(ctx, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])
        host_w.append(hw); keep.append(dw); pis.append((pi_pos, pi_vals, blinders))
        preps.append(ctx.prepare_dev(dw[0].data_ptr(), dw[1].data_ptr(), dw[2].data_ptr(), gates, table, pi_pos, pi_vals,
                                     blinders))
    return dict(evals=evals, commits=commits, gates=gates, table=table, host_w=host_w, preps=preps, pis=pis,
                pi_map0=dict(circs[0]["pi"]), keep=keep, witness=None,
                describe="full prove, synthetic withdraw-shaped circuit, %s, n=2^%d, TABLE_SIZE=1024, 7 public inputs"
                         % (args.curve, log_n))


def withdraw_workload(z, torch, ctx, dev, fld, args, log_n):
    """The reference's WithdrawCircuit filling 2^log_n rows, ONE circuit and TWO wallets / withdrawals.  Per witness the
    host uploads the variables that are not Poseidon gate outputs, the device makes the rest (PoseidonGadget.fill); the
    prover is handed the composer's own layout: the variable map and the three wire -> variable index vectors, all in HBM
    (prove.rs:49-55 wire_evals runs on the device inside the timed proof)."""
    import random
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import withdraw_workload as WW
    p = fld["r"]
    width, inputs, height = WITHDRAW_SHAPES[log_n]
    hs = WW.reference_hasher(p, width) if args.curve == "bn254" else WW.synthetic_hasher(p, width)
    insts = [WW.make_instance(hs, inputs, height, seed=0x5EED + k) for k in range(2)]
    lays = [WW.layout(hs, inst) for inst in insts]
    L0 = lays[0]
    assert lays[1].q == L0.q and lays[1].w == L0.w and sorted(lays[1].pi) == sorted(L0.pi) and lays[1].values != L0.values
    gates, n_vars = L0.n_gates, len(L0.values)
    assert gates <= (1 << log_n)
    sel = WW.setup_vectors(L0, log_n, fld["gen"])
    evals = {name: fr_to_mont_gpu(ctx, fld, sel[name]) for name in z.PK_ORDER}
    del sel
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    table = fr_to_mont_gpu(ctx, fld, insts[0]["ident_set"])
    to_mont = lambda vals: fr_to_mont_gpu(ctx, fld, vals)
    gadget = z.PoseidonGadget(ctx, hs.width, hs.half_full, hs.partial, to_mont(hs.rc), to_mont([x for row in hs.mds for x in row]),
                              to_mont([hs.tag])[0])
    assert gadget.vars_per_hash == hs.per_hash
    for base, ins in L0.hash_calls:
        gadget.hash(base, ins)
    gadget.stage()
    idx = [np.asarray(w, dtype=np.uint32) for w in L0.w]
    d_idx = [torch.from_numpy(x.view(np.int32)).to(dev) for x in idx]
    rnd = random.Random(99)
    host_w, keep, preps, pis, dev_ms = [], [], [], [], []
    for lay in lays:
        host_vals = to_mont(lay.values)                        # zeros where the device writes
        d_vars = torch.from_numpy(host_vals.view(np.int64)).to(dev)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(5):                                     # the same values every time: min of 5
            t = time.perf_counter()
            gadget.fill(d_vars.data_ptr(), n_vars, check=False)
            ctx.synchronize()
            ts.append(1e3 * (time.perf_counter() - t))
        gadget.fill(d_vars.data_ptr(), n_vars, check=True)
        dev_ms.append(min(ts))
        # wire evaluations on the host as well (the cold leg, the CPU baseline and the sharded leg take a, b, c)
        full = np.concatenate([d_vars.cpu().numpy().view(np.uint64).reshape(n_vars, 4), np.zeros((1, 4), np.uint64)])
        hw = [full[np.where(x == WW.ZERO, n_vars, x)] for x in idx]
        pi_pos = sorted(lay.pi)
        pi_vals = to_mont([lay.pi[k] for k in pi_pos])
        blinders = to_mont([rnd.randrange(p) for _ in range(z.NUM_BLINDERS)])
        host_w.append(hw); keep.append(d_vars); pis.append((pi_pos, pi_vals, blinders))
        preps.append(ctx.prepare_vars_dev(d_vars.data_ptr(), n_vars, d_idx[0].data_ptr(), d_idx[1].data_ptr(), d_idx[2].data_ptr(),
                                          gates, table, pi_pos, pi_vals, blinders))
    n_host = n_vars - len(L0.hash_calls) * hs.per_hash
    witness = {"hashes": len(L0.hash_calls), "vars_per_hash": hs.per_hash, "variables": n_vars, "host_made_variables": n_host,
               "device_made_variables": n_vars - n_host, "device_ms": round(min(dev_ms), 3),
               "launches": len(gadget._staged), "hashes_per_launch": [c for _, c, _, _ in gadget._staged],
               "is": "zkt_poseidon_gadget_witness_dev: every variable the PlonkSpecRef gadget allocates (x^2, x^4, x^5 per s-box, the W^2 "
                     "running MDS sums per round) for all the circuit's %d hashes, written into the variable map in HBM; W^2 lanes per "
                     "hash (k_poseidon_gadget_lanes: the batch is far too small for one thread per hash), one launch per dependency "
                     "level (the leaf hashes take a commitment hash); wall clock incl. launches, min of 5; outside the timed region, as "
                     "circuit.synthesize is outside proof_system::prove" % len(L0.hash_calls)}
    return dict(evals=evals, commits=commits, gates=gates, table=table, host_w=host_w, preps=preps, pis=pis,
                pi_map0=dict(L0.pi), keep=(keep, d_idx, gadget), witness=witness, oracle_twin=(hs, insts[0]),
                describe="full prove, WithdrawCircuit INPUTS=%d HEIGHT=%d Poseidon x%d (%d gates, %d Poseidon gadgets), %s, n=2^%d, "
                         "TABLE_SIZE=1024, %d public inputs" % (inputs, height, width, gates, len(L0.hash_calls), args.curve, log_n,
                                                                len(L0.pi)))


def launch_workers(args, argv):
    """`--gpus N` without a torchrun environment: start the N workers ourselves, as child processes, before this
    process has touched a GPU (it never does: no torch / HIP call happens on this path)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bn254", choices=sorted(FIELDS))
    ap.add_argument("--workload", default="withdraw", choices=["withdraw", "synthetic"],
                    help="withdraw: the reference's WithdrawCircuit sized to fill 2^log_n rows (log-n 14, 18, 19, 20, 22), "
                         "Poseidon witness made on the device; synthetic: random withdraw-shaped rows of any size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the cold / unchained single-proof legs")
    ap.add_argument("--inflight", type=int, default=0,
                    help="extra leg (1 GPU): throughput with this many independent contexts in flight on the GPU, each on its own "
                         "stream with its own SRS table, circuit and witnesses, driven by one host thread each (reported beside "
                         "the headline, which stays the single-context figure)")
    ap.add_argument("--shard", default="both", choices=["proofs", "proof", "both"],
                    help="N > 1: 'proofs' = independent proofs sharded across the GPUs (weak scaling, the headline); "
                         "'proof' = ONE proof's MSMs and 4n-coset work sharded across the GPUs (strong scaling, value = "
                         "1 / latency); 'both' (default) = the headline plus the single-proof leg in the same line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args, sys.argv[1:]))

    import torch
    import zkt_plonk_amd as z

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # one process per GPU over RCCL; ZKT_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
        backend = os.environ.get("ZKT_DIST_BACKEND", "nccl")
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    fld = FIELDS[args.curve]
    log_n, n = args.log_n, 1 << args.log_n
    t0 = time.time()
    ctx = z.Context(args.curve, dev.index)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    tau = 0x5EED5EED1234567890ABCDEF % fld["r"]
    ctx.srs_generate(tau, n + 8)

    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if args.workload == "withdraw" and log_n not in WITHDRAW_SHAPES:
        sys.stderr.write("bench.py: no withdraw shape fills n = 2^%d (have %s); using --workload synthetic\n"
                         % (log_n, sorted(WITHDRAW_SHAPES)))
        args.workload = "synthetic"
    build = withdraw_workload if args.workload == "withdraw" else synthetic_workload
    wl = build(z, torch, ctx, dev, fld, args, log_n)
    evals, gates, table, host_w, preps, pis, pi_map0 = (wl[k] for k in ("evals", "gates", "table", "host_w", "preps", "pis", "pi_map0"))
    commits = wl["commits"]
    evals_keep = evals if (world > 1 and args.shard != "proofs") else None
    if not want_cpu:
        wl["evals"] = None
        del evals
    L = fld["fq_limbs"]
    rinv_q = pow(1 << (64 * L), -1, fld["q"])
    vk = {}
    for name in z.PK_ORDER:
        xy, inf = commits[name]
        if inf:
            vk[name] = None
        else:
            x = sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv_q % fld["q"]
            y = sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv_q % fld["q"]
            vk[name] = (x, y)
    setup_s = time.time() - t0

    # Timed regime: the two witnesses alternate; every proof announces the next one (zkt_prove_set_next), as a proving
    # service with a queue would: rounds 1 and 2 of proof i+1 are issued behind the last commitments of proof i.
    chain = os.environ.get("ZKT_BENCH_NO_CHAIN") is None

    def transcript():
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
        return z.seed_transcript(tr, n, vk)                                # plonk.rs:105-106

    count = [0]

    def one_proof():
        k = count[0] & 1
        count[0] += 1
        return ctx.prove_prepared(preps[k], transcript(), preps[k ^ 1] if chain else None)

    from zkt_plonk_amd import parallel as par

    def barrier():
        par.barrier(dist)
        torch.cuda.synchronize(dev)

    proof = None
    for _ in range(args.warmup):
        proof = one_proof()
    barrier()
    # The dominant kernel is timed live over the timed region with HIP events on the stream it runs on (profile level 2:
    # one event pair per MSM).  Every other scope (~80 more event records per proof, each a few microseconds of stream
    # time) is measured in a separate, untimed pass of the same chained workload right after.
    ctx.profile_enable(2 if os.environ.get("ZKT_BENCH_NO_EVENTS") is None else 0)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        proof = one_proof()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    barrier()
    prof_acc = ctx.profile_get("msm_accumulate")
    prof_acc_launches = ctx.profile_get("msm_accumulate#launches")[0]
    prof_idle = ctx.profile_get("host_wait")
    scope_names = ("msm_accumulate", "msm_main", "msm_lag_accumulate", "msm_lag_main", "msm_fold", "msm_tail", "ntt_%d" % log_n, "ntt_%d" % (log_n + 2), "quotient",
                   "round1", "round2", "round3", "round4", "round5")
    prof_steps = max(2, min(args.steps, 6))
    ctx.profile_enable(1)
    for _ in range(prof_steps):
        one_proof()
    torch.cuda.synchronize(dev)
    prof = {k: ctx.profile_get(k) for k in scope_names}
    ctx.profile_enable(0)
    if prof_acc[0]:
        prof["msm_accumulate"] = prof_acc          # the roofline's kernel time is the timed region's own
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else None
    elapsed = par.max_over_ranks(dist, elapsed, red_dev)   # whole-job time = slowest rank
    assert proof is not None and len(proof) == (802 if args.curve == "bn254" else 1010)

    total_proofs = args.steps * world
    value = total_proofs / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- the other end of the range, same line: single proofs nobody announced -------------------------------------
    latency = None
    if rank == 0 and not args.no_latency:
        ctx.prove_prepared(preps[0], transcript())      # drains the last announcement

        def timed(fn, reps=3):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                fn()
                torch.cuda.synchronize(dev)
                ts.append(1e3 * (time.perf_counter() - t))
            return ts

        warm = timed(lambda: ctx.prove_prepared(preps[0], transcript()))
        # cold: witness in HOST memory (crosses PCIe inside the call), the lookup table handed over in a different
        # order each time (same set: nothing of the table polynomial can be reused), no announcement
        tswap = table.copy()
        tswap[[0, 1]] = tswap[[1, 0]]
        tabs = [tswap, table]
        k = [0]

        def cold():
            pi_pos, pi_vals, blinders = pis[0]
            hw = host_w[0]
            ctx.prove(hw[0], hw[1], hw[2], tabs[k[0] & 1], pi_pos, pi_vals, blinders, transcript())
            k[0] += 1

        cold_ts = timed(cold)
        latency = {"cold_single_proof_ms": round(min(cold_ts), 3),
                   "cold_is": "host witness pointers (3 x %d MiB over PCIe inside the call), fresh lookup table, no announcement; "
                              "min of 3" % (gates * 32 >> 20),
                   "unchained_single_proof_ms": round(min(warm), 3),
                   "unchained_is": "witness in HBM, table cached, no announcement; min of 3"}

    # ---- several proofs in flight: F contexts, F host threads, one GPU ---------------------------------------------------
    inflight = None
    if rank == 0 and world == 1 and args.inflight > 1:
        import threading
        workers = [(ctx, preps)]
        keep_alive = []
        for _ in range(args.inflight - 1):
            # zkt_ctx_fork: the SRS / Lagrange / circuit / twiddle tables of the first context are shared, every fork owns its
            # stream, work buffers and MSM slots; the witnesses (variable maps in HBM) are read-only and shared as well
            cx = ctx.fork()
            st = torch.cuda.Stream(dev)
            cx.set_stream(st.cuda_stream)
            workers.append((cx, preps))
            keep_alive.append(st)
        per = max(2, args.steps // args.inflight)

        def drive(cx, pp, k):
            for i in range(k):
                cx.prove_prepared(pp[i & 1], transcript(), pp[(i & 1) ^ 1] if chain else None)

        for cx, pp in workers:
            drive(cx, pp, 2)
        torch.cuda.synchronize(dev)
        t_if = time.perf_counter()
        ths = [threading.Thread(target=drive, args=(cx, pp, per)) for cx, pp in workers]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize(dev)
        dt_if = time.perf_counter() - t_if
        inflight = {"contexts": args.inflight, "proofs": per * args.inflight, "proofs_per_s": round(per * args.inflight / dt_if, 4),
                    "ms_per_proof": round(1e3 * dt_if / (per * args.inflight), 3),
                    "is": "the same chained workload on %d contexts, one host thread each: the first and its forks (zkt_ctx_fork: "
                          "shared SRS / Lagrange / circuit / twiddle tables, own stream, work buffers and MSM slots); wall clock "
                          "over all of them" % args.inflight}
        for cx, _ in workers[1:]:
            cx.close()
        del keep_alive

    # ---- the verifier the reference's caller runs after every proof (bin/src/main.rs:298; proof.rs:285-503): host C++,
    # ---- pairings included (zkt_verify), on the proof just made -----------------------------------------------------
    verify = None
    if rank == 0:
        from zkt_plonk_amd import _lib as zl
        pi_pos, pi_vals, _bl = pis[0]
        vproof = ctx.prove_prepared(preps[0], transcript())
        w_n = pow(fld["gen"], (fld["r"] - 1) >> log_n, fld["r"])
        roots = fr_to_mont_gpu(ctx, fld, [pow(w_n, i, fld["r"]) for i in pi_pos])
        vk_xy = np.stack([np.zeros(2 * L, dtype=np.uint64) if commits[name][1] else np.asarray(commits[name][0], dtype=np.uint64)
                          for name in z.PK_ORDER])
        vk_inf = [bool(commits[name][1]) for name in z.PK_ORDER]
        g_xy = ctx.srs_download(0, 1)[0]
        h2, bh2 = zl.srs_generate_g2(args.curve, tau)

        def run_verify(raw):
            t = time.perf_counter()
            ok = zl.verify(args.curve, n, vk_xy, vk_inf, roots, pi_vals, raw, g_xy, h2, bh2, transcript())
            return ok, 1e3 * (time.perf_counter() - t)

        runs = [run_verify(vproof) for _ in range(7)]
        bad = bytearray(vproof)
        bad[-40] ^= 1

        def run_batch(raws):   # zkt_verify_batch: one pairing product for the whole batch
            items = [(n, vk_xy, vk_inf, roots, pi_vals, raw, g_xy, transcript()) for raw in raws]
            t = time.perf_counter()
            ok = zl.verify_batch(args.curve, items, h2, bh2)
            return ok, 1e3 * (time.perf_counter() - t) / len(raws)

        batch_runs = [run_batch([vproof] * 16) for _ in range(3)]
        verify = {"ms": round(sorted(r[1] for r in runs)[len(runs) // 2], 3), "accepted": all(r[0] for r in runs),
                  "tampered_rejected": not run_verify(bytes(bad))[0],
                  "batch16_ms_per_proof": round(min(r[1] for r in batch_runs), 3), "batch16_accepted": all(r[0] for r in batch_runs),
                  "batch16_with_one_tampered_rejected": not run_batch([vproof] * 7 + [bytes(bad)] + [vproof] * 8)[0],
                  "is": "zkt_verify on the GPU proof: deserialisation with subgroup checks, transcript, two short G1 "
                        "multi-scalar multiplications, one folded product of two optimal-ate pairings; one host core, "
                        "median of 7"}

    # ---- roofline of the dominant kernel (MSM bucket accumulation), live HIP-event timing ----
    acc_calls, acc_ms = prof["msm_accumulate"]
    msm_points = n + 3                                              # typical MSM length of the prover
    alg_bytes = msm_points * (32 + fld["point_bytes"])              # n * (32 + 64|96) per MSM
    avg_acc_s = (acc_ms / max(acc_calls, 1)) * 1e-3
    achieved = alg_bytes / avg_acc_s / 1e9 if avg_acc_s > 0 else 0.0
    info = ctx.msm_info()
    windows = info["windows"]
    mixed_adds = windows * msm_points                               # point additions the kernel really performs
    c_ref = 3 if msm_points < 32 else (msm_points.bit_length() - 1) * 69 // 100 + 2
    w_ref = -(-fld["lam"] // c_ref)
    ref_adds = w_ref * msm_points + 2 * w_ref * ((1 << c_ref) - 1)  # reference-window formula
    msm_calls, msm_ms = prof["msm_main"]            # digits + sort + accumulate (main stream)
    tail_calls, tail_ms = prof["msm_tail"]          # bucket reduction down to the partial sums the host finishes (side stream)
    fold_calls, fold_ms = prof["msm_fold"]          # bucket fold (side stream)
    avg_msm_s = ((msm_ms + tail_ms + fold_ms) / max(msm_calls, 1)) * 1e-3
    # One mixed addition (ecx.hpp xx_add_mixed, inlined) = 6 products + 2 squarings + 1 double product over
    # L 29-bit limbs: 6 * 2L^2 + 2 * (L(L+1)/2 + L^2) + 3L^2 v_mad_u64_u32 (1467 for L = 9; the ISA has 1468).
    Lq = -(-32 * 2 * L // 29)                                      # 9 (BN254 Fq), 14 (BLS12-381 Fq)
    mads_per_add = 6 * 2 * Lq * Lq + 2 * (Lq * (Lq + 1) // 2 + Lq * Lq) + 3 * Lq * Lq
    mad_ceiling = N_SIMD * 64 * CLOCK_HZ / MAD_CYCLES               # v_mad_u64_u32 issue ceiling, lanes/s
    traffic, traffic_src = pmc_traffic("k_msm_accumulate", args.curve, log_n, args.workload)
    valu_busy, valu_src = pmc_valu_busy("k_msm_accumulate", args.curve, log_n, args.workload)
    roofline = {
        "kernel": "k_msm_accumulate", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_ms": round(avg_acc_s * 1e3, 4), "launches": acc_calls,
        "kernel_launches": prof_acc_launches,
        "launches_are": "MSMs: the dense commitments only (coefficient-form MSMs of n + 2 .. n + 3 scalars), avg_launch_ms per "
                        "MSM.  The three commitments of a round that exist together (a b c; q_lo q_mid q_hi) go out as ONE "
                        "kernel launch with blockIdx.y = MSM, which rocprof lists as one dispatch of three times the work: "
                        "`kernel_launches` dispatches covered `launches` MSMs.  The Lagrange-basis commitments of t / h1 / h2 / "
                        "z2 run the same kernel over a few thousand pairs and are listed under `commitments`",
        "frac_of_mad_issue_ceiling": round(mads_per_add * mixed_adds / avg_acc_s / mad_ceiling, 4) if avg_acc_s > 0 else None,
        "frac_of_mad_issue_ceiling_at_residency": round(mads_per_add * mixed_adds / avg_acc_s / (mad_ceiling * MAD_CYCLES / MAD_CYCLES_RESIDENT), 4) if avg_acc_s > 0 else None,
        "valu_busy": valu_busy, "valu_busy_source": valu_src,
        "note": "integer-ALU bound (v_mad_u64_u32 issue), not HBM bound: the contract's HBM fraction is reported, the "
                "binding ceiling is frac_of_mad_issue_ceiling (see int_alu); traffic = measured gather traffic of W*n "
                "random table points, not re-reads",
    }
    int_alu = {
        "msm_g1_adds_per_s_reference_formula": round(ref_adds / avg_msm_s, 1) if avg_msm_s > 0 else None,
        "msm_mixed_adds_per_s_accumulate": round(mixed_adds / avg_acc_s, 1) if avg_acc_s > 0 else None,
        "mads_per_mixed_add": mads_per_add,
        "accumulate_mad_per_s": round(mads_per_add * mixed_adds / avg_acc_s, 1) if avg_acc_s > 0 else None,
        "mad_issue_ceiling_per_s": round(mad_ceiling, 1),
        "frac_of_mad_issue_ceiling": round(mads_per_add * mixed_adds / avg_acc_s / mad_ceiling, 4) if avg_acc_s > 0 else None,
        "msm_avg_ms": round(avg_msm_s * 1e3, 4), "msm_launches": msm_calls,
        "msm_main_stream_avg_ms": round(msm_ms / max(msm_calls, 1), 4),
        "msm_tail_avg_ms": round((tail_ms + fold_ms) / max(tail_calls, 1), 4),
        "frac_of_mad_issue_ceiling_at_residency": roofline["frac_of_mad_issue_ceiling_at_residency"],
        "ceilings": "frac_of_mad_issue_ceiling prices a multiply-add at %.1f cycles (the best measured issue rate, the r01-r03 "
                    "definition); _at_residency at %.1f (the rate at the kernel's 3 waves per SIMD, r04's definition)" % (MAD_CYCLES, MAD_CYCLES_RESIDENT),
    }
    lag_calls, lag_ms = prof["msm_lag_main"]
    lag_acc_calls, lag_acc_ms = prof["msm_lag_accumulate"]
    lag_info = ctx.lagrange_info()
    per_proof = lambda calls: round(calls / float(prof_steps), 2)
    commitments = {
        "dense_msms_per_proof": per_proof(msm_calls), "lagrange_msms_per_proof": per_proof(lag_calls),
        "lagrange_main_stream_avg_ms": round(lag_ms / lag_calls, 4) if lag_calls else None,
        "lagrange_accumulate_avg_ms": round(lag_acc_ms / lag_acc_calls, 4) if lag_acc_calls else None,
        "lagrange_table_bases": lag_info["bases"],
        "is": "h1, h2, z2 (and t when the table changes) are piecewise constant as evaluation vectors; they are committed in "
              "the Lagrange basis of the domain, where the MSM's scalars are the differences of neighbouring evaluations "
              "(include/zkt_plonk.h zkt_commit_evals_dev): the same points, hence the same proof bytes, as the "
              "reference's coefficient-form commitments (prove.rs:166-180,249-251); a circuit with dense lookups gains "
              "nothing and loses nothing",
    }
    # NTT: HBM fraction (the BASELINE metric) and the fraction of the mad ceiling.  Products per transform: one twiddle
    # product per butterfly output that has a non-unit twiddle, (N/2) log2 N at most; L^2-term schoolbook + reduction
    # = 2 L^2 + L multiply-adds each on L = 9 limbs.
    Lr = 9
    mads_per_mul = 2 * Lr * Lr + Lr
    ntt = {}
    for lg in (log_n, log_n + 2):
        calls, ms = prof["ntt_%d" % lg]
        if calls:
            avg = ms / calls * 1e-3
            muls = (1 << lg) // 2 * lg
            ntt["ntt_2^%d" % lg] = {"avg_ms": round(avg * 1e3, 4), "launches": calls,
                                    "GB/s": round(64.0 * (1 << lg) / avg / 1e9, 2),
                                    "frac_hbm": round(64.0 * (1 << lg) / avg / 1e9 / HBM_PEAK_GBS, 5),
                                    "frac_of_mad_issue_ceiling_upper": round(muls * mads_per_mul / avg / mad_ceiling, 4)}
    qc, qms = prof["quotient"]
    if qc:
        qavg = qms / qc * 1e-3
        ntt["quotient"] = {"avg_ms": round(qavg * 1e3, 4), "GB/s": round(23 * 32 * 4 * n / qavg / 1e9, 2),
                           "frac_hbm": round(23 * 32 * 4 * n / qavg / 1e9 / HBM_PEAK_GBS, 5)}

    out = {
        "metric": "proofs/sec (withdraw, n=2^%d)" % log_n, "value": round(value, 4), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (256-bit Montgomery)",
        "data": "synthetic",
        "config": {"workload": wl["describe"], "parallelism": "proofs sharded across %d GPU(s)" % world,
                   "chained": bool(chain), "distinct_witnesses": 2, "witness_on_device": True, "table_cached": True,
                   "lagrange_commits": lag_info["log_n"] == log_n,
                   "proof_bytes": len(proof), "setup_s": round(setup_s, 1)},
        "roofline": roofline, "int_alu": int_alu, "commitments": commitments, "kernels": ntt,
        "rounds_ms": {k: round(prof[k][1] / prof[k][0], 3) for k in ("round1", "round2", "round3", "round4", "round5") if prof[k][0]},
        "kernels_measured": "msm_accumulate: HIP events inside the timed region; the other scopes and rounds_ms: %d further "
                            "chained proofs with every scope on, outside the timed region (stream time, first to last "
                            "launch of the scope; round4 / round5 contain the early round1 / round2 of the next proof)" % prof_steps,
    }
    if prof_idle[0]:
        idle_ms = prof_idle[1] / args.steps
        out["gpu_active"] = {
            "main_stream_busy_ms_per_proof": round(ms_per_step - idle_ms, 3), "main_stream_idle_ms_per_proof": round(idle_ms, 3),
            "busy_frac": round(1.0 - idle_ms / ms_per_step, 4), "host_round_trips_per_proof": round(prof_idle[0] / args.steps, 2),
            "is": "HIP events on the proving stream inside the timed region: idle = from the moment the stream drains while the "
                  "host waits for a round's commitments / evaluations until the next launch (zkt_profile_get \"host_wait\"); "
                  "busy = wall clock per proof minus that.  The bucket-reduction tails run on a side stream during part of "
                  "the idle time",
        }
    if wl.get("witness") is not None:
        out["witness"] = wl["witness"]
        out["poseidon_witness_ms"] = wl["witness"]["device_ms"]
    if latency is not None:
        out["latency"] = latency
        # the stricter regimes as first-class figures beside `value` (proofs/s of back-to-back single proofs)
        out["value_unchained"] = round(1e3 / latency["unchained_single_proof_ms"], 4)
        out["value_cold"] = round(1e3 / latency["cold_single_proof_ms"], 4)
    if inflight is not None:
        out["inflight"] = inflight
    if verify is not None:
        out["verify_ms"] = verify["ms"]
        out["verify"] = verify

    # ---- ONE proof across all the GPUs (SURVEY.md 8e / BASELINE.json configs[4]) -------------------------------------
    if world > 1 and args.shard != "proofs":
        refs = [ctx.prove_prepared(preps[k], transcript()) for k in range(2)]        # this GPU alone, for the bytes
        progress = {"phase": "start"}
        guard = leg_watchdog(rank, out, float(os.environ.get("ZKT_SHARD_LEG_TIMEOUT", "180")), progress)
        try:
            sh = sharded_leg(z, par, dist, dev, args, fld, tau, evals_keep, vk, host_w, table, pis, gates, refs, barrier,
                             progress)
        except Exception as e:      # reported, never fatal for the headline
            sh = {"error": "%s: %s" % (type(e).__name__, e)}
        guard.cancel()
        out["single_proof_sharded"] = sh
        if args.shard == "proof" and "ms_per_proof" in sh:
            out.update(value=round(1e3 / sh["ms_per_proof"], 4), ms_per_step=sh["ms_per_proof"], scaling="strong")
            out["config"]["parallelism"] = "one proof sharded across %d GPU(s)" % world

    if want_cpu:
        pi_pos, pi_vals, blinders = pis[0]
        gpu_proof = ctx.prove_prepared(preps[0], transcript())
        out["cpu_baseline"] = cpu_baseline(ctx, args.curve, log_n, evals, host_w[0], table, pi_map0, blinders, vk, gpu_proof,
                                           wl.get("oracle_twin"))
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    # a sharded proof whose bytes differ from the single-GPU proof's is a correctness failure of the multi-GPU transport
    # (the RCCL device path cannot be rehearsed on one-GPU boxes): the line above is printed, the run fails
    sh = out.get("single_proof_sharded")
    if isinstance(sh, dict) and sh.get("proof_bytes_equal_single_gpu") is False:
        sys.stderr.write("bench.py rank %d: sharded proof bytes differ from the single-GPU proof\n" % rank)
        sys.exit(4)


def leg_watchdog(rank, out, seconds, progress):
    """The single-proof leg is an extra: if a rank is still inside it after `seconds` (a collective waiting for a peer that
    failed, a hung GPU), every rank's own timer ends its process.  Rank 0 first prints the headline line measured before the
    leg, with the leg's last completed phase and collective count in its error record; every other rank waits a few seconds
    for that line to get out.  All exit 0: the headline is valid and the abandoned leg is in the line (and on stderr) -- a
    non-zero exit would make the launcher discard a good measurement over an extra."""
    import threading

    def give_up():
        where = "last phase '%s', %d collective calls entered" % (progress.get("phase", "?"), progress.get("calls", lambda: -1)())
        if rank == 0:
            o = dict(out)
            o["single_proof_sharded"] = {"error": "no result within %.0f s (%s); leg abandoned, the headline is unaffected"
                                                  % (seconds, where)}
            print(json.dumps(o), flush=True)
        else:
            time.sleep(5.0)
        sys.stderr.write("bench.py rank %d: single-proof leg abandoned after %.0f s (%s)\n" % (rank, seconds, where))
        sys.stderr.flush()
        os._exit(0)

    t = threading.Timer(seconds, give_up)
    t.daemon = True
    t.start()
    return t


def sharded_leg(z, par, dist, dev, args, fld, tau, evals, vk, host_w, table, pis, gates, refs, barrier, progress=None):
    """ONE proof on all the GPUs of the job: every commitment is an index-range-sharded MSM (each rank keeps 1 / N of
    the SRS, one all-gather of partial sums per prover round), the nine 4n-coset transforms and the quotient pass run on
    the rank's class of the coset with no exchange, one all-gather (4n x 32 B in total) precedes the inverse transform;
    the rest is replicated.  Same two witnesses, chained, as the headline; every rank checks its bytes against the
    proofs it made alone."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    log_n, n = args.log_n, 1 << args.log_n
    ctx = z.Context(args.curve, dev.index)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    comm = par.TorchComm(dist, dev)
    ctx.set_comm(comm)
    progress = progress if progress is not None else {}
    progress["calls"] = lambda: comm.calls
    progress["phase"] = "communicator attached"
    lo, hi = par.shard_range(n + 8, rank, world)
    ctx.srs_generate_slice(tau, lo, hi - lo, n + 8)
    progress["phase"] = "SRS slice generated"
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    progress["phase"] = "sharded setup done"
    L = fld["fq_limbs"]
    preps = []
    for hw, (pi_pos, pi_vals, blinders) in zip(host_w, pis):
        dw = [torch.from_numpy(x.view(np.int64)).to(dev) for x in hw]
        preps.append((ctx.prepare_dev(dw[0].data_ptr(), dw[1].data_ptr(), dw[2].data_ptr(), gates, table, pi_pos, pi_vals,
                                      blinders), dw))

    def transcript():
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
        return z.seed_transcript(tr, n, vk)

    count = [0]

    def one():
        k = count[0] & 1
        count[0] += 1
        return k, ctx.prove_prepared(preps[k][0], transcript(), preps[k ^ 1][0])

    ok = True
    for i in range(max(args.warmup, 2)):
        k, pr = one()
        ok = ok and pr == refs[k]
        progress["phase"] = "warm-up proof %d done" % i
    barrier()
    calls0, bytes0 = ctx.comm_stats()
    t0 = time.perf_counter()
    for i in range(args.steps):
        k, pr = one()
        progress["phase"] = "timed proof %d done" % i
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    barrier()
    progress["phase"] = "timed region done"
    ok = ok and pr == refs[k]
    calls1, bytes1 = ctx.comm_stats()
    red_dev = dev if dist.get_backend() == "nccl" else None
    dt = par.max_over_ranks(dist, dt, red_dev)
    all_ok = par.max_over_ranks(dist, 0.0 if ok else 1.0, red_dev) == 0.0
    info = ctx.msm_info()
    ctx.close()
    return {"ms_per_proof": round(1e3 * dt / args.steps, 3), "proofs_per_s": round(args.steps / dt, 4),
            "proof_bytes_equal_single_gpu": bool(all_ok), "srs_points_per_gpu": hi - lo,
            "msm_window_bits": info["window_bits"], "coset_points_per_gpu": 4 * n // world,
            "collectives_per_proof": (calls1 - calls0) / args.steps,
            "bytes_sent_per_gpu_per_proof": (bytes1 - bytes0) // args.steps, "backend": dist.get_backend()}


def csrc_digest():
    """sha256 over the library's sources (zkt-plonk_amd/csrc/*): what a committed counter profile is valid for.  The GPU
    box has no .git, so staleness is decided on the sources themselves, not on commits."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zkt-plonk_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc", ".cpp", ".h")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_files(prefix, curve, log_n, workload):
    """Committed counter summaries (tools/pmc_traffic.sh / pmc_valu.sh write a first line
    `# curve=... log_n=... workload=... csrc_digest=...`) that were taken on THIS configuration, newest first, each with
    the digest of the sources it was taken on."""
    import glob
    import re
    out = []
    for f in glob.glob(os.path.join(ROOT, "profiles", prefix + "_r*.txt")):
        with open(f) as fh:
            head = fh.readline()
        m = dict(re.findall(r"(\w+)=(\S+)", head)) if head.startswith("#") else {}
        if m.get("curve") != curve or m.get("log_n") != str(log_n) or m.get("workload", workload) != workload:
            continue
        key = [int(x) for x in re.findall(r"\d+", os.path.basename(f))]
        out.append((key, f, m.get("csrc_digest")))
    return [(f, dg) for _, f, dg in sorted(out, reverse=True)]


def _static_source(f, dg):
    return {"file": "profiles/" + os.path.basename(f), "static": True, "csrc_digest": dg,
            "is": "rocprofv3 --pmc pass of this bench configuration on these very sources, committed; not measured in this run"}


def pmc_traffic(kernel, curve, log_n, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh ->
    profiles/pmc_traffic_*.txt: FETCH_SIZE and WRITE_SIZE in separate runs, KiB per launch; on gfx950 FETCH_SIZE counts
    64-B units for the 128-B requests of wide coalesced loads, MI355X_MICROARCH.md).  The accumulation kernel's reads are
    scattered 64-B points, so its raw FETCH_SIZE is taken as is.  A profile counts only if it was taken on this curve,
    size and workload AND on the sources the library is built from now: otherwise (None, reason)."""
    cur = csrc_digest()
    files = _pmc_files("pmc_traffic", curve, log_n, workload)
    for f, dg in files:
        if dg != cur:
            continue
        for line in open(f):
            if line.startswith(kernel):
                parts = line.split()
                try:
                    return round((float(parts[-2]) + float(parts[-1])) * 1024.0), _static_source(f, dg)
                except ValueError:
                    pass
    return None, {"static": True, "stale": True,
                  "is": "no committed PMC pass matches this configuration and the current sources (csrc_digest %s; have %s)"
                        % (cur, [(os.path.basename(f), dg) for f, dg in files][:3])}


def pmc_valu_busy(kernel, curve, log_n, workload):
    """VALUBusy of `kernel` from the committed rocprofv3 PMC pass (tools/pmc_valu.sh -> profiles/pmc_valu_*.txt):
    SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE per XCD) -- rocprof's own derived-metric formula, gfx950 has no
    entry of its own.  Same validity rule as pmc_traffic."""
    cur = csrc_digest()
    files = _pmc_files("pmc_valu", curve, log_n, workload)
    for f, dg in files:
        if dg != cur:
            continue
        c, vals = None, {}
        for line in open(f):
            if line.startswith("#"):
                continue
            if not line.startswith(" "):
                c = line.split()[0] if line.strip() else None
                continue
            if c is not None and c.startswith(kernel):
                parts = line.split()
                if len(parts) == 2 and parts[0] in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):
                    vals[parts[0]] = float(parts[1])
        if len(vals) == 2 and vals["GRBM_GUI_ACTIVE"] > 0:
            return round(vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * vals["GRBM_GUI_ACTIVE"] / 8.0), 4), _static_source(f, dg)
    return None, {"static": True, "stale": True,
                  "is": "no committed PMC pass matches this configuration and the current sources (csrc_digest %s)" % cur}


def cpu_baseline(ctx, curve, log_n, evals, wires, table, pi, blinders, vk_pts, gpu_proof, oracle_twin=None):
    """ONE full proof of the same workload by the CPU oracle (oracle/fastplonk.py: the array twin of the restated
    prove.rs:59-470, every O(n) loop in oracle/coracle.cpp -- a port of ark-poly's radix-2 FFT, ark-ec's Pippenger
    and the prover's own loops, OpenMP over the host cores), timed, on the same SRS / witness / blinders; its bytes
    must equal the GPU proof's.  Key preparation (ExtendedProverKey) is outside the timed region on both sides."""
    from oracle import coracle as K, fields as F, fastplonk as FP, plonk as P
    cv = F.CURVES[curve]
    n = 1 << log_n
    srs = ctx.srs_download(0, n + 8)
    keys = FP.setup(cv, srs, log_n, evals, commitments=False)
    keys.commits = dict(vk_pts)
    vk = keys.verifier_key(cv, pi.keys())
    bl = K.fr_from_mont(cv, blinders)
    t = time.perf_counter()
    cpu_proof = FP.prove(cv, srs, keys, wires[0], wires[1], wires[2], table, pi, P.new_seeded_transcript(cv, vk), bl)
    dt = time.perf_counter() - t
    res = {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": K.num_threads(), "kind": "port",
           "sample": "1 full proof (n=2^%d, same SRS / witness / blinders as the GPU proof) in %.2f s" % (log_n, dt),
           "proof_bytes_equal_gpu": cpu_proof == gpu_proof}
    if oracle_twin is not None:
        # the checker's own synthesis of the same withdrawal (oracle/composer.py: the reference's composer restated gate by
        # gate): its wire values -- Poseidon variables included -- must be the ones the device made (untimed)
        from oracle import composer as OC
        hs, inst = oracle_twin
        prm = OC.PoseidonParams(cv.fr.p, hs.width, hs.half_full, hs.partial, hs.rc, hs.mds, hs.tag)
        cs = OC.Composer(cv, inst["ident_set"], 1024)
        OC.withdraw_synthesize(cs, prm, inst["secrets"], inst["identifiers"], inst["amounts"], inst["poes"], inst["root"],
                               inst["new_secret"], inst["new_identifier"], inst["withdraw_amount"])
        a, b, c = cs.wire_evals(cs.n_gates)
        res["witness_equals_oracle_composer"] = bool(
            cs.check_satisfied() and all(np.array_equal(K.fr_to_mont(cv, x), np.asarray(w)) for x, w in zip((a, b, c), wires)))
    return res


if __name__ == "__main__":
    main()
