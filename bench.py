#!/usr/bin/env python3
"""bench.py -- proofs/sec of the MI355X prover hot path on a synthetic withdraw-shaped circuit.

A "step" is one full pass of proof_system::prove (plonk-core/src/proof_system/prove.rs:59-470) over one
batch of synthetic witness rows already resident in HBM: 9 iNTT(n), 10 NTT(4n), the fused quotient
pass, two grand products, 13 KZG MSMs of ~n points, the 12 evaluations and the two openings, with
Fiat-Shamir (Merlin) on the host.  Default workload: BASELINE.json configs[3] shape -- BN254,
n = 2^20 rows, TABLE_SIZE = 1024, 7 public inputs.

With N > 1 (one process per GPU, launched by torch.distributed.run) independent proofs are sharded
across the ranks (no data-path collective): scaling = "weak".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the MSM bucket accumulation),
measured live with HIP events on the stream the kernel runs on; `cpu_baseline` times the CPU oracle
(a port, oracle/coracle.cpp) on a bounded sample of the same workload on this box's host cores.
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
MAD_CYCLES = 4.4               # measured v_mad_u64_u32 cycles / wave-instr / SIMD (profiles/microbench_r01.txt)
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


def synthetic_circuit(fld, log_n, table_size=1024, n_public=7, seed=0x5EED):
    """Withdraw-shaped synthetic trace of exactly 2^log_n rows: mul / add / linear
    gates chained through copy constraints (output of row i-1 = left input of row i), a lookup row every 16,
    n_public public-input rows.  Pure Python integers; no reference or oracle code involved."""
    import random
    rnd = random.Random(seed)
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
    for i in range(gates - n_public):
        if i % 16 == 15:
            t = table[rr(len(table))]
            a[i] = t; c[i] = t
            q_l[i] = 1; q_o[i] = p - 1; q_lk[i] = 1
        else:
            if prev_out is None:
                a[i] = rr(p)
            else:
                a[i] = c[prev_out]
                # one permutation cycle {Output(prev_out), Left(i)}
                s3[prev_out] = roots[i]
                s1[i] = K2 * roots[prev_out] % p
            b[i] = rr(p)
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
        v = rr(p)
        c[i] = v
        q_o[i] = p - 1
        pi[i] = v
    q_table = [0] * table_size + [1] * (n - table_size)
    sel = dict(q_m=q_m, q_l=q_l, q_r=q_r, q_o=q_o, q_c=q_c, sigma1=s1, sigma2=s2, sigma3=s3, q_lookup=q_lk,
               q_table=q_table)
    return dict(n=n, gates=gates, a=a, b=b, c=c, sel=sel, table=table, pi=pi)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bn254", choices=sorted(FIELDS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    ctx.srs_generate(0x5EED5EED1234567890ABCDEF % fld["r"], n + 8)

    circ = synthetic_circuit(fld, log_n)
    # proof_system::setup on the device (setup.rs:42-166): selector / sigma / table-mask evaluations -> ProverKey,
    # ExtendedProverKey and the ten VerifierKey commitments that seed the transcript
    evals = {name: fr_to_mont_gpu(ctx, fld, circ["sel"][name]) for name in z.PK_ORDER}
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
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
    gates = circ["gates"]
    wires = [torch.from_numpy(fr_to_mont_gpu(ctx, fld, circ[k][:gates]).view(np.int64)).to(dev) for k in "abc"]
    table = fr_to_mont_gpu(ctx, fld, circ["table"])
    pi_pos = sorted(circ["pi"])
    pi_vals = fr_to_mont_gpu(ctx, fld, [circ["pi"][k] for k in pi_pos])
    import random
    rnd = random.Random(99)
    blinders = fr_to_mont_gpu(ctx, fld, [rnd.randrange(fld["r"]) for _ in range(z.NUM_BLINDERS)])
    del circ
    setup_s = time.time() - t0

    # The same witness is proved over and over; every proof announces the next one (zkt_prove_set_next), as a proving
    # service with a queue would: rounds 1 and 2 of proof i+1 are issued behind the last commitments of proof i.
    prep = ctx.prepare_dev(wires[0].data_ptr(), wires[1].data_ptr(), wires[2].data_ptr(), gates, table, pi_pos, pi_vals,
                           blinders)
    chain = os.environ.get("ZKT_BENCH_NO_CHAIN") is None

    def one_proof():
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=fld["lam"], fq_bytes=8 * L)
        z.seed_transcript(tr, n, vk)                                # plonk.rs:105-106
        return ctx.prove_prepared(prep, tr, prep if chain else None)

    from zkt_plonk_amd import parallel as par

    def barrier():
        par.barrier(dist)
        torch.cuda.synchronize(dev)

    proof = None
    for _ in range(args.warmup):
        proof = one_proof()
    barrier()
    ctx.profile_enable(os.environ.get("ZKT_BENCH_NO_EVENTS") is None)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        proof = one_proof()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    barrier()
    prof = {k: ctx.profile_get(k) for k in ("msm_accumulate", "msm_main", "msm_tail", "ntt_%d" % log_n, "ntt_%d" % (log_n + 2),
                                            "quotient")}
    ctx.profile_enable(False)
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else None
    elapsed = par.max_over_ranks(dist, elapsed, red_dev)   # whole-job time = slowest rank
    assert proof is not None and len(proof) == (802 if args.curve == "bn254" else 1010)

    total_proofs = args.steps * world
    value = total_proofs / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

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
    msm_calls, msm_ms = prof["msm_main"]            # digits + sort + accumulate + bucket fold (main stream)
    tail_calls, tail_ms = prof["msm_tail"]          # bucket reduction (side stream, overlaps the next MSM)
    avg_msm_s = ((msm_ms + tail_ms) / max(msm_calls, 1)) * 1e-3
    # One mixed addition (ecx.hpp xx_add_mixed, inlined) = 6 products + 2 squarings + 1 double product over
    # L 29-bit limbs: 6 * 2L^2 + 2 * (L(L+1)/2 + L^2) + 3L^2 v_mad_u64_u32 (1467 for L = 9; the ISA has 1468).
    Lq = -(-32 * 2 * L // 29)                                      # 9 (BN254 Fq), 14 (BLS12-381 Fq)
    mads_per_add = 6 * 2 * Lq * Lq + 2 * (Lq * (Lq + 1) // 2 + Lq * Lq) + 3 * Lq * Lq
    mad_ceiling = N_SIMD * 64 * CLOCK_HZ / MAD_CYCLES               # v_mad_u64_u32 issue ceiling, lanes/s
    roofline = {
        "kernel": "k_msm_accumulate", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
        "avg_launch_ms": round(avg_acc_s * 1e3, 4), "launches": acc_calls,
        "note": "integer-ALU bound (v_mad_u64_u32 issue), not HBM bound: see int_alu",
    }
    int_alu = {
        "msm_g1_adds_per_s_reference_formula": round(ref_adds / avg_msm_s, 1) if avg_msm_s > 0 else None,
        "msm_mixed_adds_per_s_accumulate": round(mixed_adds / avg_acc_s, 1) if avg_acc_s > 0 else None,
        "mads_per_mixed_add": mads_per_add,
        "accumulate_mad_per_s": round(mads_per_add * mixed_adds / avg_acc_s, 1) if avg_acc_s > 0 else None,
        "mad_issue_ceiling_per_s": round(mad_ceiling, 1),
        "frac_of_mad_issue_ceiling": round(mads_per_add * mixed_adds / avg_acc_s / mad_ceiling, 4) if avg_acc_s > 0 else None,
        "msm_avg_ms": round(avg_msm_s * 1e3, 4), "msm_launches": msm_calls,
        "msm_tail_avg_ms": round(tail_ms / max(tail_calls, 1), 4),
    }
    ntt = {}
    for lg in (log_n, log_n + 2):
        calls, ms = prof["ntt_%d" % lg]
        if calls:
            avg = ms / calls * 1e-3
            ntt["ntt_2^%d" % lg] = {"avg_ms": round(avg * 1e3, 4), "launches": calls,
                                    "GB/s": round(64.0 * (1 << lg) / avg / 1e9, 2),
                                    "frac_hbm": round(64.0 * (1 << lg) / avg / 1e9 / HBM_PEAK_GBS, 5)}
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
        "config": {"workload": "full prove, synthetic withdraw-shaped circuit, %s, n=2^%d, TABLE_SIZE=1024, 7 public inputs"
                               % (args.curve, log_n), "parallelism": "proofs sharded across %d GPU(s)" % world,
                   "chained": bool(chain),
                   "proof_bytes": len(proof), "setup_s": round(setup_s, 1)},
        "roofline": roofline, "int_alu": int_alu, "kernels": ntt,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctx, args.curve, log_n)
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(ctx, curve, log_n):
    """The CPU oracle (oracle/coracle.cpp: a port of ark-poly's radix-2 FFT and ark-ec's Pippenger, OpenMP
    over the host cores) timed on one MSM(n), one iNTT(n) and one coset-NTT(4n); a proof is priced as
    14 MSM + 9 iNTT(n) + 10 NTT(4n), the pointwise passes being left out."""
    from oracle import coracle as K, fields as F
    cv = F.CURVES[curve]
    n = 1 << log_n
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 1 << 61, size=(n, 4), dtype=np.uint64)
    srs = ctx.srs_download(0, n)
    t = time.perf_counter(); K.msm_mont(cv, srs, sc, True); t_msm = time.perf_counter() - t
    t = time.perf_counter(); K.ntt_mont(cv, log_n, True, False, sc); t_intt = time.perf_counter() - t
    t = time.perf_counter(); K.ntt_mont(cv, log_n + 2, False, True, sc); t_ntt4 = time.perf_counter() - t
    per_proof = 14 * t_msm + 9 * t_intt + 10 * t_ntt4   # the reference's own counts
    return {"value": round(1.0 / per_proof, 5), "unit": "proofs/s", "cores": K.num_threads(), "kind": "port",
            "sample": "1 MSM(2^%d) %.2fs + 1 iNTT(2^%d) %.2fs + 1 coset-NTT(2^%d) %.2fs, scaled to 14/9/10 per proof"
                      % (log_n, t_msm, log_n, t_intt, log_n + 2, t_ntt4)}


if __name__ == "__main__":
    main()
