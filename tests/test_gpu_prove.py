"""GPU parity of the full prover (proof bytes) against the CPU oracle and the committed golden proof."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, curve as C, plonk as P, coracle as K
from helpers import field_elems, unhex_point, run_sharded_ranks, sharded_exchange_bytes

CURVES = [F.BN254, F.BLS12_381]


@pytest.fixture(scope="module")
def ctxs():
    import zkt_plonk_amd as z
    c = {cv.name: z.Context(cv.name, 0) for cv in CURVES}
    yield c
    for x in c.values():
        x.close()


def _gpu_prove(z, ctx, cv, cs, pk, vk, srs_arr, blinders, kind="merlin"):
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    ctx.srs_load(srs_arr[:n + 8])
    prover = z.GpuProver(ctx, log_n, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64)
                                      for k in z.PK_ORDER})
    tr = z.Transcript(kind, "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
    z.seed_transcript(tr, vk.n, vk.commits)
    a, b, c = cs.wire_evals(cs.n_gates)
    pi = {pos: K.fr_to_mont(cv, [v])[0] for pos, v in cs.pi.items()}
    return prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c),
                        K.fr_to_mont(cv, cs.table) if cs.table else np.zeros((0, 4), dtype=np.uint64),
                        pi, K.fr_to_mont(cv, blinders), tr)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_reference_test_circuit_golden_proof_bytes(cv, ctxs, golden):
    """plonk-core/src/plonk.rs:144-218 TestCircuit, n = 128: the GPU proof equals the committed bytes."""
    import zkt_plonk_amd as z
    g = golden[cv.name]["test_circuit"]
    tau = int(golden[cv.name]["tau"], 16)
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    assert {k: unhex_point(v) for k, v in g["vk"].items()} == vk.commits
    blinders = field_elems(cv.fr.p, g["blinder_seed"], P.NUM_BLINDERS)
    proof = _gpu_prove(z, ctxs[cv.name], cv, cs, pk, vk, srs_arr, blinders)
    assert len(proof) == (802 if cv.name == "bn254" else 1010)
    assert proof.hex() == g["proof_bytes"]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("gates,table_size", [(60, 16), (1000, 64), (4000, 1024)])
def test_synthetic_circuits_match_oracle(cv, gates, table_size, ctxs):
    import zkt_plonk_amd as z
    cs = P.synthetic_circuit(cv, gates, table_size, seed=gates)
    assert cs.check_satisfied()
    n = cs.circuit_bound()
    tau = 0xABCDEF0123456789 + gates
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 77 + gates, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders)
    got = _gpu_prove(z, ctxs[cv.name], cv, cs, pk, vk, srs_arr, blinders)
    assert got == want.serialize(cv)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    assert P.verify(cv, tau, vk, want, P.new_seeded_transcript(cv, vk), pis)


@pytest.mark.parametrize("n_public", [0, 1, 2, 15, 16, 17, 40])
def test_public_input_counts(n_public, ctxs):
    """Up to 16 public inputs the quotient kernel evaluates PI(X) from rotations of the l1 coset; beyond that
    the polynomial goes through iNTT + coset NTT as in the reference (prove.rs:258-262).  Both must agree with
    the oracle byte for byte, including the empty and the boundary cases."""
    import zkt_plonk_amd as z
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 200, 16, seed=900 + n_public, n_public=n_public)
    assert cs.check_satisfied()
    assert len(cs.pi) <= n_public
    n = cs.circuit_bound()
    tau = 0x5151 + n_public
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 4000 + n_public, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders)
    got = _gpu_prove(z, ctxs[cv.name], cv, cs, pk, vk, srs_arr, blinders)
    assert got == want.serialize(cv)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_setup_on_the_device(cv, ctxs):
    """zkt_circuit_setup = proof_system::setup (setup.rs:42-166): from the SetupComposer's evaluation vectors to the
    VerifierKey commitments and a loaded prover; both against the oracle's setup + prove."""
    import zkt_plonk_amd as z
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 700, 32, seed=4242)
    n = cs.circuit_bound()
    tau = 0x7E57
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    evals = P.setup_evals(be, cs)
    ctx.srs_load(srs_arr)
    prover, commits = z.GpuProver.setup(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, evals[k]) for k in z.PK_ORDER})
    q = cv.fq.p
    rinv = pow(1 << (64 * cv.fq.limbs64), -1, q)
    L = cv.fq.limbs64
    for name in z.PK_ORDER:
        xy, inf = commits[name]
        want = vk.commits[name]
        if want is None:
            assert inf
            continue
        x = sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv % q
        y = sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv % q
        assert not inf and (x, y) == want, name
    blinders = field_elems(cv.fr.p, 31, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders)
    a, b, c = cs.wire_evals(cs.n_gates)
    tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=8 * L)
    z.seed_transcript(tr, n, vk.commits)
    got = prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                       {i: K.fr_to_mont(cv, [v])[0] for i, v in cs.pi.items()}, K.fr_to_mont(cv, blinders), tr)
    assert got == want.serialize(cv)


def test_witness_gather_on_the_device(ctxs):
    """prove.rs:49-55 wire_evals on the device: the composer's variable map + per-gate index vectors give the same
    proof as the three evaluation vectors; an index outside the map is an error."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 500, 32, seed=55)
    n = cs.circuit_bound()
    tau = 0xABBA
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 9, P.NUM_BLINDERS)
    want = _gpu_prove(z, ctx, cv, cs, pk, vk, srs_arr, blinders)
    assert want == P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    to_idx = lambda ws: np.array([0xFFFFFFFF if v == P.ZERO_VAR else v for v in ws], dtype=np.uint32)
    pi_pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos])

    def run(w_l):
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        z.seed_transcript(tr, vk.n, vk.commits)
        return ctx.prove_vars(K.fr_to_mont(cv, cs.values), w_l, to_idx(cs.w_r), to_idx(cs.w_o), K.fr_to_mont(cv, cs.table),
                              pi_pos, pi_vals, K.fr_to_mont(cv, blinders), tr)

    assert run(to_idx(cs.w_l)) == want
    bad = to_idx(cs.w_l)
    bad[3] = len(cs.values)          # one past the last variable
    with pytest.raises(z.ZktError) as e:
        run(bad)
    assert e.value.code == 1


def _bench_workload(z, ctx, cv, log_n, tau):
    """bench.py's workload (BASELINE.json configs[3] shape: TABLE_SIZE 1024, 7 public inputs, 2^log_n rows) set up on
    the device; returns what both sides need: Montgomery arrays, the SRS the GPU generated, the GPU-made VerifierKey."""
    import bench as B
    n = 1 << log_n
    ctx.srs_generate(tau, n + 8)
    circ = B.synthetic_circuit(B.FIELDS[cv.name], log_n)
    evals = {name: K.fr_to_mont(cv, circ["sel"][name]) for name in z.PK_ORDER}
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    L = cv.fq.limbs64
    q = cv.fq.p
    rinv = pow(1 << (64 * L), -1, q)
    pts = {}
    for name in z.PK_ORDER:
        xy, inf = commits[name]
        pts[name] = None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv % q,
                                      sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv % q)
    gates = circ["gates"]
    pi_pos = sorted(circ["pi"])
    w = dict(a=K.fr_to_mont(cv, circ["a"][:gates]), b=K.fr_to_mont(cv, circ["b"][:gates]),
             c=K.fr_to_mont(cv, circ["c"][:gates]), table=K.fr_to_mont(cv, circ["table"]), pi=circ["pi"], pi_pos=pi_pos,
             pi_vals=K.fr_to_mont(cv, [circ["pi"][k] for k in pi_pos]))
    vk = P.VerifierKey(n, [pow(cv.fr.root_of_unity(n), i, cv.fr.p) for i in pi_pos], pts)
    return evals, w, vk



def _check_srs_ends(cv, tau, srs):
    """The device-generated SRS against the oracle at BOTH ends: the first 64 powers by the oracle's generator, the last 64
    as [tau^i mod r] G by double-and-add on Python integers (a sampled head alone would miss a drift along the key)."""
    from oracle import curve as C
    count = srs.shape[0]
    assert np.array_equal(srs[:64], K.srs_mont(cv, tau, 64))
    G = C.generator(cv)
    want = [C.scalar_mul(cv, pow(tau, i, cv.fr.p), G) for i in range(count - 64, count)]
    assert K.points_from_mont(cv, srs[count - 64:]) == want


def _gpu_prove_arrays(z, ctx, cv, w, vk, blinders):
    tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=8 * cv.fq.limbs64)
    z.seed_transcript(tr, vk.n, vk.commits)
    return ctx.prove(w["a"], w["b"], w["c"], w["table"], w["pi_pos"], w["pi_vals"], K.fr_to_mont(cv, blinders), tr)


@pytest.mark.parametrize("cvname,log_n", [("bn254", 14), ("bls12_381", 14), ("bn254", 17), ("bls12_381", 17), ("bn254", 18),
                                          ("bls12_381", 18), ("bn254", 19), ("bn254", 20), ("bls12_381", 20)])
def test_headline_configs_proof_bytes_equal_cpu_oracle(cvname, log_n, ctxs):
    """BASELINE.json configs[0] (BN254, n = 2^14) and configs[3] (BN254, n = 2^20: the workload bench.py times), plus
    BLS12-381 at 2^14 and 2^20 and the sizes between (2^17, 2^18: the reference CLI's default feature set, 2^19: each has
    its own MSM digit width and launch regime): the GPU proof equals, byte for byte, the proof of the CPU oracle's array prover
    (oracle/fastplonk.py, pinned to the big-integer restatement of prove.rs:59-470 in tests/test_coracle.py) on the
    same SRS, witness, public inputs and blinders.  VerifierKey commitments: all ten at 2^14, two at 2^20."""
    import zkt_plonk_amd as z
    from oracle import fastplonk as FP
    cv = F.CURVES[cvname]
    ctx = ctxs[cv.name]
    n = 1 << log_n
    tau = 0x5EED5EED1234567890ABCDEF % cv.fr.p
    evals, w, vk = _bench_workload(z, ctx, cv, log_n, tau)
    srs = ctx.srs_download(0, n + 8)
    _check_srs_ends(cv, tau, srs)
    keys = FP.setup(cv, srs, log_n, evals, commitments=log_n <= 14)
    if log_n <= 14:
        assert keys.commits == vk.commits
    else:
        for name in ("q_c", "sigma3"):
            assert FP.commit(cv, srs, keys.pk[name]) == vk.commits[name], name
    blinders = field_elems(cv.fr.p, 2020 + log_n, P.NUM_BLINDERS)
    got = _gpu_prove_arrays(z, ctx, cv, w, vk, blinders)
    want = FP.prove(cv, srs, keys, w["a"], w["b"], w["c"], w["table"], w["pi"], P.new_seeded_transcript(cv, vk), blinders)
    assert len(got) == (802 if cv.name == "bn254" else 1010)
    assert got == want
    if cv.name == "bn254" and log_n == 14:     # the committed fixture of configs[0] (tests/golden/vectors_r02.json)
        import hashlib, json, os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r02.json")) as f:
            e = json.load(f)["config0_bn254_2_14"]
        assert e["blinder_seed"] == 2020 + log_n and int(e["tau"], 16) == tau
        assert hashlib.sha256(got).hexdigest() == e["proof_sha256"]
    pis = [w["pi"][k] for k in w["pi_pos"]]
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), pis)
    if cv.name == "bn254" and log_n == 20:
        # configs[3]'s proof once more as ONE proof across 2 / 4 / 8 ranks (thread ranks on this GPU: SRS slices, sharded
        # setup, class transforms of 2^21 / 2^20 / 2^19 points, the quotient exchange): every rank's bytes == the CPU
        # oracle's.  7 public inputs: evaluated from l1 rotations at world 2 and 4, through the transform at world 8.
        _sharded_legs(z, cv, n, srs, evals, vk, w, blinders, want, (2, 4, 8))


def _sharded_legs(z, cv, n, srs, evals, vk, w, blinders, want, worlds):
    job = (w["a"], w["b"], w["c"], w["table"], w["pi_pos"], w["pi_vals"], K.fr_to_mont(cv, blinders))
    for world in worlds:
        out = run_sharded_ranks(z, cv, n, srs, {"evals": evals}, vk, [job], world)
        for r in range(world):
            proofs, (calls, sent), (setup_calls, setup_sent) = out[r]
            assert proofs == [want], (world, r)
            assert calls - setup_calls == 5
            assert sent - setup_sent == sharded_exchange_bytes(cv, n, world, 1), (world, r)


def test_sharded_proof_with_many_public_inputs_at_2_20(ctxs):
    """More than QUOTIENT_PI_DIRECT_MAX (16) public inputs at the headline size: PI(X) goes through the inverse transform
    and the class transform on every world size (prover.hip `pi_direct` false), single GPU and 4 ranks, against the CPU
    oracle's array prover on the same SRS, witness and blinders."""
    import bench as B
    import zkt_plonk_amd as z
    from oracle import fastplonk as FP
    cv = F.BN254
    ctx = ctxs[cv.name]
    log_n, n = 20, 1 << 20
    tau = 0x5EED5EED1234567890ABCDEF % cv.fr.p
    ctx.srs_generate(tau, n + 8)
    circ = B.synthetic_circuit(B.FIELDS[cv.name], log_n, n_public=20, seed=0xB16)
    evals = {name: K.fr_to_mont(cv, circ["sel"][name]) for name in z.PK_ORDER}
    srs = ctx.srs_download(0, n + 8)
    keys = FP.setup(cv, srs, log_n, evals, commitments=True)
    vk = keys.verifier_key(cv, circ["pi"].keys())
    gates = circ["gates"]
    pi_pos = sorted(circ["pi"])
    assert len(pi_pos) == 20
    w = dict(a=K.fr_to_mont(cv, circ["a"][:gates]), b=K.fr_to_mont(cv, circ["b"][:gates]),
             c=K.fr_to_mont(cv, circ["c"][:gates]), table=K.fr_to_mont(cv, circ["table"]), pi=circ["pi"], pi_pos=pi_pos,
             pi_vals=K.fr_to_mont(cv, [circ["pi"][k] for k in pi_pos]))
    blinders = field_elems(cv.fr.p, 4040, P.NUM_BLINDERS)
    want = FP.prove(cv, srs, keys, w["a"], w["b"], w["c"], w["table"], w["pi"], P.new_seeded_transcript(cv, vk), blinders)
    z.GpuProver.setup(ctx, log_n, evals)
    assert _gpu_prove_arrays(z, ctx, cv, w, vk, blinders) == want
    _sharded_legs(z, cv, n, srs, evals, vk, w, blinders, want, (4,))


def test_config4_bls12_381_2_22_proof_bytes_equal_cpu_oracle(ctxs):
    """BASELINE.json configs[4], single-GPU leg: BLS12-381 at n = 2^22.  The CPU oracle's array prover needs about a minute
    for this proof on the GPU box's 16 threads; the GPU proof must equal it byte for byte, the oracle's verifier
    (proof.rs:285-503, pairing replaced by the trapdoor identity) accepts it under the GPU-made VerifierKey, one of whose
    commitments is checked against the CPU port, and a flipped evaluation is rejected."""
    import zkt_plonk_amd as z
    from oracle import fastplonk as FP
    cv = F.BLS12_381
    ctx = ctxs[cv.name]
    log_n, n = 22, 1 << 22
    tau = 0x5EED5EED1234567890ABCDEF % cv.fr.p
    evals, w, vk = _bench_workload(z, ctx, cv, log_n, tau)
    srs = ctx.srs_download(0, n + 8)
    _check_srs_ends(cv, tau, srs)
    keys = FP.setup(cv, srs, log_n, evals, commitments=False)
    assert FP.commit(cv, srs, keys.pk["q_c"]) == vk.commits["q_c"]
    blinders = field_elems(cv.fr.p, 2020, P.NUM_BLINDERS)
    proof = _gpu_prove_arrays(z, ctx, cv, w, vk, blinders)
    assert len(proof) == 1010
    want = FP.prove(cv, srs, keys, w["a"], w["b"], w["c"], w["table"], w["pi"], P.new_seeded_transcript(cv, vk), blinders)
    del keys
    assert proof == want
    pis = [w["pi"][k] for k in w["pi_pos"]]
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, proof), P.new_seeded_transcript(cv, vk), pis)
    bad = bytearray(proof)
    bad[-40] ^= 1                                   # inside the last evaluation (h2_eval)
    assert not P.verify(cv, tau, vk, P.proof_deserialize(cv, bytes(bad)), P.new_seeded_transcript(cv, vk), pis)
    # free the 2^22 circuit (tens of GiB of HBM) before the ranks below (and the next test) load their own
    ctx.srs_generate(tau, 64)
    z.GpuProver(ctx, 3, {k: np.zeros((0, 4), dtype=np.uint64) for k in z.PK_ORDER})
    # configs[4] as BASELINE.json names it: ONE proof across 8 ranks (thread ranks on this GPU, ~9 GiB of HBM each):
    # SRS slices of 2^19 powers, sharded setup, class transforms of 2^21 points on the class and on the class of
    # "omega-next" ((r + 4) mod 8), public inputs through the transform, 64 MiB of quotient exchange per rank.
    _sharded_legs(z, cv, n, srs, evals, vk, w, blinders, want, (8,))


def test_chained_proofs_with_prefetch(ctxs):
    """zkt_prove_set_next: rounds 1 and 2 of the announced proof are issued behind the current proof's last commitments.
    The bytes must not change - when the announcement is honoured, when a different proof follows, and when another MSM
    call gets in between (which invalidates the early work)."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 900, 64, seed=77)
    n = cs.circuit_bound()
    tau = 0xC4A1
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx.srs_load(srs_arr)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    a, b, c = cs.wire_evals(cs.n_gates)
    wires = []
    for w in (a, b, c):                       # the witness resident in HBM, through the library's own memory calls
        arr = K.fr_to_mont(cv, w)
        d = ctx.alloc(arr.nbytes)
        ctx.upload(d, arr)
        wires.append(d)
    pi_pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos])
    table = K.fr_to_mont(cv, cs.table)
    bl = [field_elems(cv.fr.p, 500 + i, P.NUM_BLINDERS) for i in range(4)]
    want = [P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), x).serialize(cv) for x in bl]
    preps = [ctx.prepare_dev(wires[0], wires[1], wires[2], cs.n_gates, table, pi_pos, pi_vals, K.fr_to_mont(cv, x))
             for x in bl]

    def tr():
        t = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        return z.seed_transcript(t, vk.n, vk.commits)

    # honoured announcements: 0 -> 1 -> 2
    assert ctx.prove_prepared(preps[0], tr(), preps[1]) == want[0]
    assert ctx.prove_prepared(preps[1], tr(), preps[2]) == want[1]
    # 2 was announced and runs; it announces 3, but 0 follows instead: the early work is redone
    assert ctx.prove_prepared(preps[2], tr(), preps[3]) == want[2]
    assert ctx.prove_prepared(preps[0], tr(), preps[1]) == want[0]
    # an unrelated MSM between the announcement and the proof invalidates the early commitments
    ctx.msm(K.fr_to_mont(cv, [3, 5, 7]))
    assert ctx.prove_prepared(preps[1], tr()) == want[1]
    # and a plain proof afterwards is unaffected
    assert ctx.prove_prepared(preps[3], tr()) == want[3]
    for d in wires:
        ctx.free(d)


def test_chained_proofs_with_distinct_witnesses_tables_and_public_inputs(ctxs):
    """The alternate work set of zkt_prove_set_next (prover.hip swap_work_sets): proofs i and i+1 differ in everything a
    prover is handed per proof -- wire values, lookup table (so the early round 2 rebuilds the table polynomial while
    proof i still opens its own), public inputs, blinders -- and arrive in all three input forms (device wires, host
    wires, the composer's variables + index vectors, whose host staging borrows the quotient vector and the round-5
    work buffer).  Every proof must equal the oracle's bytes whether announced, unannounced or announced and dropped."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    css = [P.synthetic_circuit(cv, 900, 64, seed=78, value_seed=100 + k) for k in range(4)]
    css.append(css[1])                       # same table as proof 1 but reached from a different one
    cs0 = css[0]
    n = cs0.circuit_bound()
    tau = 0xC4A2
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs0, True)
    for cs in css[1:]:
        assert cs.check_satisfied() and P.setup_evals(be, cs) == P.setup_evals(be, cs0)   # one circuit, many witnesses
        assert cs.table != cs0.table or cs is css[0]
    assert len({tuple(sorted(cs.pi.items())) for cs in css[:4]}) == 4
    ctx.srs_load(srs_arr)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    bl = [field_elems(cv.fr.p, 700 + i, P.NUM_BLINDERS) for i in range(len(css))]
    want = [P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), x).serialize(cv)
            for cs, x in zip(css, bl)]
    assert len(set(want)) == len(want)
    to_idx = lambda ws: np.array([0xFFFFFFFF if v == P.ZERO_VAR else v for v in ws], dtype=np.uint32)
    dev_bufs = []
    preps = []
    for k, (cs, x) in enumerate(zip(css, bl)):
        pi_pos = sorted(cs.pi)
        pi_vals = K.fr_to_mont(cv, [cs.pi[i] for i in pi_pos])
        table = K.fr_to_mont(cv, cs.table)
        blm = K.fr_to_mont(cv, x)
        a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
        if k % 3 == 0:       # witness resident in HBM
            ds = []
            for arr in (a, b, c):
                d = ctx.alloc(arr.nbytes)
                ctx.upload(d, arr)
                ds.append(d)
            dev_bufs += ds
            preps.append(ctx.prepare_dev(ds[0], ds[1], ds[2], cs.n_gates, table, pi_pos, pi_vals, blm))
        elif k % 3 == 1:     # host wire vectors
            preps.append(ctx.prepare_host(a, b, c, table, pi_pos, pi_vals, blm))
        else:                # the composer's layout, staged through the quotient vector
            preps.append(ctx.prepare_vars(K.fr_to_mont(cv, cs.values), to_idx(cs.w_l), to_idx(cs.w_r), to_idx(cs.w_o),
                                          table, pi_pos, pi_vals, blm))

    def tr():
        t = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        return z.seed_transcript(t, vk.n, vk.commits)

    # honoured announcements across all input forms and changing tables: 0 -> 1 -> 2 -> 3 -> 4(table of 1) -> 0
    order = [0, 1, 2, 3, 4, 0]
    for i, k in enumerate(order):
        nxt = preps[order[i + 1]] if i + 1 < len(order) else None
        assert ctx.prove_prepared(preps[k], tr(), nxt) == want[k], (i, k)
    # announced and dropped: 1 announces 2, but 3 arrives; then unannounced proofs in reverse order
    assert ctx.prove_prepared(preps[1], tr(), preps[2]) == want[1]
    assert ctx.prove_prepared(preps[3], tr(), preps[0]) == want[3]
    for k in (2, 1, 0):
        assert ctx.prove_prepared(preps[k], tr()) == want[k], k
    for d in dev_bufs:
        ctx.free(d)


def test_failed_proof_withdraws_the_announcement(ctxs):
    """A proof that fails (a looked-up value outside the table) must leave no announcement armed: the successor's
    buffers may be gone by the next call.  The next plain proof equals the oracle's bytes."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 300, 32, seed=91, value_seed=1)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0xFA11, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx.srs_load(srs_arr)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    blinders = field_elems(cv.fr.p, 3, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    pi_pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[i] for i in pi_pos])
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    good = ctx.prepare_host(a, b, c, K.fr_to_mont(cv, cs.table), pi_pos, pi_vals, K.fr_to_mont(cv, blinders))
    row = next(i for i, q in enumerate(cs.q_lookup) if q)
    cbad = c.copy()
    cbad[row] = K.fr_to_mont(cv, [(cs.value_of(cs.w_o[row]) + 1) % cv.fr.p])[0]
    bad = ctx.prepare_host(a, b, cbad, K.fr_to_mont(cv, cs.table), pi_pos, pi_vals, K.fr_to_mont(cv, blinders))
    # the announced successor lives in arrays that are released right after the failure
    doomed = ctx.prepare_host(a.copy(), b.copy(), c.copy(), K.fr_to_mont(cv, cs.table), pi_pos, pi_vals.copy(),
                              K.fr_to_mont(cv, blinders))

    def tr():
        t = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        return z.seed_transcript(t, vk.n, vk.commits)

    import ctypes
    L = z.lib()

    def raw_prove(prep):   # zkt_prove alone: no zkt_prove_set_next call that would overwrite a stale announcement
        t = tr()
        out = (ctypes.c_uint8 * 2048)()
        ln = ctypes.c_size_t(0)
        rc = L.zkt_prove(ctx.handle, ctypes.byref(prep.struct), t.handle, out, 2048, ctypes.byref(ln))
        return rc, bytes(out[:ln.value])

    ctx.check(L.zkt_prove_set_next(ctx.handle, ctypes.byref(doomed.struct)))
    rc, _ = raw_prove(bad)
    assert rc == 8
    for arr in doomed._keep:
        if isinstance(arr, np.ndarray):
            arr[...] = 0xDEADBEEF            # whoever still reads these produces garbage
    ctx.profile_enable(True)
    for rep in range(2):
        before = ctx.profile_get("msm_main")[0]
        rc, got = raw_prove(good)
        assert rc == 0 and got == want
        # 13 MSMs (12 once the table commitment is cached); early rounds of a stale successor would add 3 to 6
        assert ctx.profile_get("msm_main")[0] - before <= 13, rep
    ctx.profile_enable(False)
    del doomed


def test_srs_reload_invalidates_the_cached_table_commitment(ctxs):
    """circuit_load -> prove -> srs_load(another key) -> prove(same table): the table polynomial's cached commitment
    belongs to the first key and must not reach the second proof (the circuit state itself is SRS independent)."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 250, 32, seed=23)
    n = cs.circuit_bound()
    a, b, c = cs.wire_evals(cs.n_gates)
    pi = {pos: K.fr_to_mont(cv, [v])[0] for pos, v in cs.pi.items()}
    blinders = field_elems(cv.fr.p, 8, P.NUM_BLINDERS)
    loaded = False
    for tau in (1111, 2222, 1111):
        srs_arr = K.srs_mont(cv, tau, n + 8)
        be = K.CBackend(cv, srs_arr)
        pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
        want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
        ctx.srs_load(srs_arr)
        if not loaded:      # the circuit is loaded ONCE; only the key changes afterwards
            prover = z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
            loaded = True
        for _ in range(2):  # second proof under each key takes the cached-table path
            tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
            got = prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                               pi, K.fr_to_mont(cv, blinders), tr)
            assert got == want, tau


def test_repeated_proofs_reuse_the_table_polynomial(ctxs):
    """Second and third proof on the same loaded circuit take the cached-table path (same table), then a
    different table invalidates the cache; every proof must still equal the oracle's bytes."""
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 300, 32, seed=21)
    n = cs.circuit_bound()
    tau = 31337
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx.srs_load(srs_arr)
    prover = z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    a, b, c = cs.wire_evals(cs.n_gates)
    pi = {pos: K.fr_to_mont(cv, [v])[0] for pos, v in cs.pi.items()}
    for rep, seed in enumerate((3, 4, 5)):
        blinders = field_elems(cv.fr.p, seed, P.NUM_BLINDERS)
        want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
        tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
        got = prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                           pi, K.fr_to_mont(cv, blinders), tr)
        assert got == want, rep
    # same circuit, two table entries swapped: t changes (same set), the cache must not be used
    cs2 = P.ConstraintSystem.__new__(P.ConstraintSystem)
    cs2.__dict__.update(cs.__dict__)
    cs2.table = list(cs.table)
    cs2.table[0], cs2.table[1] = cs2.table[1], cs2.table[0]
    blinders = field_elems(cv.fr.p, 6, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs2, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
    got = prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs2.table),
                       pi, K.fr_to_mont(cv, blinders), tr)
    assert got == want


def test_ethereum_transcript_proof_matches_oracle(ctxs):
    import zkt_plonk_amd as z
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 200, 32, seed=5)
    n = cs.circuit_bound()
    tau = 424242
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 9, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk, "ethereum"), blinders)
    got = _gpu_prove(z, ctxs[cv.name], cv, cs, pk, vk, srs_arr, blinders, kind="ethereum")
    assert got == want.serialize(cv)


def test_prover_error_paths(ctxs):
    import zkt_plonk_amd as z
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 100, 16, seed=8)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 777, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 1, P.NUM_BLINDERS)
    # a looked-up value that is not in the table -> Error::ElementNotIndexedInTable (multiset.rs:121)
    row = next(i for i, q in enumerate(cs.q_lookup) if q)
    bad = P.ConstraintSystem.__new__(P.ConstraintSystem)
    bad.__dict__.update(cs.__dict__)
    bad.values = list(cs.values)
    bad.values[cs.w_o[row]] = (cs.values[cs.w_o[row]] + 1) % cv.fr.p
    with pytest.raises(z.ZktError) as e:
        _gpu_prove(z, ctx, cv, bad, pk, vk, srs_arr, blinders)
    assert e.value.code == 8
    # an unsatisfied arithmetic gate -> the quotient is not a polynomial of degree 3n+5
    row = next(i for i, q in enumerate(cs.q_lookup) if not q and cs.w_o[i] >= 0 and cs.q_o[i])
    bad.values = list(cs.values)
    bad.values[cs.w_o[row]] = (cs.values[cs.w_o[row]] + 1) % cv.fr.p
    with pytest.raises(z.ZktError) as e:
        _gpu_prove(z, ctx, cv, bad, pk, vk, srs_arr, blinders)
    assert e.value.code in (8, 9)
    # SRS too short for the circuit -> kzg10 TooManyCoefficients
    ctx.srs_load(srs_arr[:n // 2])
    prover = z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
    a, b, c = cs.wire_evals(cs.n_gates)
    with pytest.raises(z.ZktError) as e:
        prover.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                     {p_: K.fr_to_mont(cv, [v])[0] for p_, v in cs.pi.items()}, K.fr_to_mont(cv, blinders), tr)
    assert e.value.code == 5


def test_foreign_transcript_callbacks(ctxs):
    """zkt_prove_with: the transcript lives on the caller's side (the Rust shim's T: TranscriptProtocol);
    here the four callbacks are Python functions driving the oracle's Merlin, fed with Montgomery limbs."""
    import ctypes
    import zkt_plonk_amd as z
    from zkt_plonk_amd._lib import ProveInputs, u64p
    cv = F.BN254
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 120, 16, seed=33)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 99991, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 12, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    ctx.srs_load(srs_arr)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})

    tr = P.new_seeded_transcript(cv, vk)   # the "foreign" transcript object
    p, q = cv.fr.p, cv.fq.p
    rinv_r, rinv_q = pow(1 << 256, -1, p), pow(1 << 256, -1, q)
    R = (1 << 256) % p

    def limbs(ptr, k):
        return sum(int(ptr[i]) << (64 * (i - 4 * k)) for i in range(4 * k, 4 * k + 4))

    U64 = ctypes.POINTER(ctypes.c_uint64)
    CB_U64 = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64)
    CB_SC = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_char_p, U64, ctypes.c_size_t, ctypes.c_int)
    CB_CM = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_char_p, U64, ctypes.c_int)
    CB_CH = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_char_p, U64)

    def cb_u64(user, label, v):
        tr.append_u64(label.decode(), v)

    def cb_sc(user, label, ptr, count, single):
        vals = [limbs(ptr, k) * rinv_r % p for k in range(count)]
        if single:
            tr.append_scalar(label.decode(), vals[0])
        else:
            tr.append_scalars(label.decode(), vals)

    def cb_cm(user, label, ptr, inf):
        pt = None if inf else (limbs(ptr, 0) * rinv_q % q, limbs(ptr, 1) * rinv_q % q)
        tr.append_commitment(label.decode(), pt)

    def cb_ch(user, label, out):
        v = tr.challenge_scalar(label.decode()) * R % p
        for i in range(4):
            out[i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF

    class VT(ctypes.Structure):
        _fields_ = [("user", ctypes.c_void_p), ("append_u64", CB_U64), ("append_scalars", CB_SC),
                    ("append_commitment", CB_CM), ("challenge_scalar", CB_CH)]

    vt = VT(None, CB_U64(cb_u64), CB_SC(cb_sc), CB_CM(cb_cm), CB_CH(cb_ch))
    a, b, c = (K.fr_to_mont(cv, x) for x in cs.wire_evals(cs.n_gates))
    table = K.fr_to_mont(cv, cs.table)
    pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[k] for k in pos])
    bl = K.fr_to_mont(cv, blinders)
    posarr = (ctypes.c_size_t * len(pos))(*pos)
    inp = ProveInputs(u64p(a), u64p(b), u64p(c), a.shape[0], u64p(table), table.shape[0], posarr, u64p(pi_vals),
                      len(pos), u64p(bl), 0)
    out = (ctypes.c_uint8 * 2048)()
    ln = ctypes.c_size_t(0)
    L = z.lib()
    L.zkt_prove_with.argtypes = [ctypes.c_void_p, ctypes.POINTER(ProveInputs), ctypes.POINTER(VT),
                                 ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    rc = L.zkt_prove_with(ctx.handle, ctypes.byref(inp), ctypes.byref(vt), out, 2048, ctypes.byref(ln))
    assert rc == 0, L.zkt_last_error(ctx.handle)
    assert bytes(out[:ln.value]) == want


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_prove_from_the_reference_cli_key_files(cv, ctxs, tmp_path):
    """a prover service started from the files the reference CLI writes (bin/src/main.rs:105-111):
    --ck -> zkt_srs_load_file, --pk -> zkt_circuit_load_file, --vk -> the transcript seed.  Files come from the oracle's
    writers ("parity unpinned": the reference holds no key file); the proof must equal the oracle's bytes."""
    import zkt_plonk_amd as z
    from zkt_plonk_amd import _lib
    from oracle import keyfile as KF
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 1500, 64, seed=404)
    n = cs.circuit_bound()
    tau = 0xF11E5
    srs_arr = K.srs_mont(cv, tau, 4 * n + 1)             # PC::trim keeps 4n + 1 powers (plonk.rs:79-85)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (4 * n + 1), cs, True)
    blinders = field_elems(cv.fr.p, 61, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (4 * n + 1), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    (tmp_path / "ck").write_bytes(KF.committer_key_bytes(cv, K.points_from_mont(cv, srs_arr)))
    (tmp_path / "pk").write_bytes(KF.prover_key_bytes(cv, pk))
    (tmp_path / "vk").write_bytes(KF.verifier_key_bytes(cv, vk))
    ctx.srs_load_file(str(tmp_path / "ck"), max_powers=n + 8)
    assert ctx.msm_info()["srs_count"] == n + 8
    ctx.circuit_load_file(str(tmp_path / "pk"), n.bit_length() - 1)
    n_, roots, commits, inf = _lib.keyfile_verifier_key(str(tmp_path / "vk"), cv.name)
    pts = {name: (None if inf[k] else K.points_from_mont(cv, commits[k:k + 1])[0]) for k, name in enumerate(z.PK_ORDER)}
    tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
    z.seed_transcript(tr, n_, pts)
    a, b, c = cs.wire_evals(cs.n_gates)
    pos = sorted(cs.pi)
    got = ctx.prove(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table), pos,
                    K.fr_to_mont(cv, [cs.pi[i] for i in pos]), K.fr_to_mont(cv, blinders), tr)
    assert got == want


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_epk_file_of_the_reference_cli_against_the_device_derivation(cv, ctxs, tmp_path):
    """SURVEY.md 8f.2, --epk (bin/src/main.rs:34-35,108-109): the library derives the ExtendedProverKey on the device
    (keys/mod.rs:78-146) and never reads the file; zkt_circuit_check_epk_file says whether a file the reference wrote holds
    the same seventeen vectors.  The file comes from the oracle's writer ("parity unpinned": the reference holds no key file)."""
    import zkt_plonk_amd as z
    from oracle import keyfile as KF
    ctx = ctxs[cv.name]
    cs = P.synthetic_circuit(cv, 1500, 64, seed=405)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0xE9C, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    (tmp_path / "pk").write_bytes(KF.prover_key_bytes(cv, pk))
    blob = KF.extended_prover_key_bytes(cv, epk)
    f = tmp_path / "epk"
    f.write_bytes(blob)
    ctx.srs_load(srs_arr)
    ctx.circuit_load_file(str(tmp_path / "pk"), n.bit_length() - 1)
    assert ctx.check_epk_file(str(f)) is None
    # one value changed in every kind of vector: a coset, an evaluation vector, x, zh, l_1 -- found where it is
    starts, off = [], 0
    vecs = KF.extended_prover_key_vectors(epk)
    for name in KF.EPK_ORDER:
        starts.append(off + 8)
        off += 8 + 32 * len(vecs[name])
    for vec, at in ((0, 0), (3, 4 * n - 1), (5, 7), (8, n - 1), (13, 70001 % (4 * n)), (14, 2), (15, 4 * n - 3), (16, 65536 % (4 * n))):
        b = bytearray(blob)
        b[starts[vec] + 32 * at] ^= 1
        f.write_bytes(bytes(b))
        assert ctx.check_epk_file(str(f)) == (vec, at)
    # the key of another circuit size: the first vector's length says so; a file that is not one: an error
    cs2 = P.synthetic_circuit(cv, 100, 16, seed=406)
    n2 = cs2.circuit_bound()
    be2 = K.CBackend(cv, srs_arr[:n2 + 8])
    pk2, epk2, vk2 = P.setup(be2, [None] * (n2 + 8), cs2, True)
    f.write_bytes(KF.extended_prover_key_bytes(cv, epk2))
    assert ctx.check_epk_file(str(f)) == (0, -1)
    f.write_bytes(blob[:-5])
    with pytest.raises(z.ZktError):
        ctx.check_epk_file(str(f))
    # the circuit still proves after the check borrowed its work buffers
    blinders = field_elems(cv.fr.p, 62, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    assert _gpu_prove(z, ctx, cv, cs, pk, vk, srs_arr, blinders) == want


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_gpu_proof_through_the_product_verifier(cv, ctxs):
    """Prover and verifier of the same library: a GPU proof goes through zkt_verify_prepare (proof.rs:285-503 without
    the pairings) and the resulting pairs satisfy L == tau W under the test trapdoor, i.e. e(L, h) == e(W, tau h)."""
    import zkt_plonk_amd as z
    from zkt_plonk_amd import _lib
    from oracle import curve as C
    cs = P.synthetic_circuit(cv, 2000, 64, seed=808)
    n = cs.circuit_bound()
    tau = 0x1234567
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    proof = _gpu_prove(z, ctxs[cv.name], cv, cs, pk, vk, srs_arr, field_elems(cv.fr.p, 5, P.NUM_BLINDERS))
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
    z.seed_transcript(tr, vk.n, vk.commits)
    pairs, inf = _lib.verify_prepare(cv.name, vk.n, K.points_to_mont(cv, [vk.commits[k] for k in z.PK_ORDER]),
                                     [vk.commits[k] is None for k in z.PK_ORDER], K.fr_to_mont(cv, vk.pi_roots),
                                     K.fr_to_mont(cv, pis), proof, srs_arr[0], tr)
    pts = [None if inf[i] else K.points_from_mont(cv, pairs[i:i + 1])[0] for i in range(4)]
    assert pts[0] == C.scalar_mul(cv, tau, pts[1]) and pts[2] == C.scalar_mul(cv, tau, pts[3])
    # ... and through the complete verifier, pairings included (zkt_verify), with SonicKZG10's h and beta h = tau h
    from oracle import pairing as PR
    from test_pairing_host import g2_mont
    T = PR.Tower(cv)
    H = PR.G2_GENERATORS[cv.name]
    commits = K.points_to_mont(cv, [vk.commits[k] for k in z.PK_ORDER])
    vinf = [vk.commits[k] is None for k in z.PK_ORDER]

    def full(raw, bh):
        t2 = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        z.seed_transcript(t2, vk.n, vk.commits)
        return _lib.verify(cv.name, vk.n, commits, vinf, K.fr_to_mont(cv, vk.pi_roots), K.fr_to_mont(cv, pis), raw, srs_arr[0],
                           g2_mont(cv, [H])[0], bh, t2)

    assert full(proof, g2_mont(cv, [T.g2_mul(tau, H)])[0])
    assert not full(proof, g2_mont(cv, [T.g2_mul(tau + 1, H)])[0])
    flipped = bytearray(proof)
    flipped[-5] ^= 2
    assert not full(bytes(flipped), g2_mont(cv, [T.g2_mul(tau, H)])[0])


def test_short_soak_of_chained_proofs():
    """tools/soak.py in its short form: 400 chained proofs at n = 2^12 over three witnesses x two table orders in device
    and host form, announcements honoured, dropped or withheld at random, an unrelated MSM now and then; every proof must
    reproduce the bytes its inputs gave unchained."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "12", "400"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "SOAK OK: 400 proofs, 0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("cvname,gates", [("bn254", 900), ("bn254", 9000), ("bls12_381", 3000)])
def test_contexts_in_flight_prove_the_same_bytes(cvname, gates):
    """Several proofs in flight on one GPU: three contexts (own stream, own copy of the key and circuit), one host thread
    each, every thread proving a chain of four different proofs of the same circuit (zkt_prove_set_next) while the others do
    the same -- the regime that hides the latency chain of a small proof (n = 2^14: 309 -> 460 proofs/s).  The kernels'
    function attributes, the Lagrange-basis table of each context and the deferred bucket reductions are all exercised
    concurrently; every proof must carry the CPU oracle's bytes."""
    import threading
    import zkt_plonk_amd as z
    cv = F.CURVES[cvname]
    cs = P.synthetic_circuit(cv, gates, 256, seed=gates + 3)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0xF11E + gates, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    pkm = {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64) for k in z.PK_ORDER}
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    pi_pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos])
    table = K.fr_to_mont(cv, cs.table)
    nctx, chain = 3, 4
    bl = [[field_elems(cv.fr.p, 9000 + 10 * t + i, P.NUM_BLINDERS) for i in range(chain)] for t in range(nctx)]
    want = [[P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), x).serialize(cv) for x in row]
            for row in bl]
    got = [[None] * chain for _ in range(nctx)]
    errors = []
    start = threading.Barrier(nctx)

    def worker(t):
        try:
            ctx = z.Context(cv.name, 0)
            ctx.srs_load(srs_arr)
            z.GpuProver(ctx, n.bit_length() - 1, pkm)
            preps = [ctx.prepare_host(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, x)) for x in bl[t]]
            start.wait(timeout=300)
            for rep in range(2):                  # the second pass runs on warm tables, fully overlapped
                for i in range(chain):
                    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8),
                                           vk.n, vk.commits)
                    got[t][i] = ctx.prove_prepared(preps[i], tr, preps[i + 1] if i + 1 < chain else None)
                    assert got[t][i] == want[t][i], (t, rep, i)
            assert ctx.lagrange_info()["log_n"] == n.bit_length() - 1
            ctx.close()
        except BaseException as e:
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nctx)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=900)
    assert not errors, errors
    assert got == want


@pytest.mark.parametrize("cvname,gates", [("bn254", 3000), ("bls12_381", 900)])
def test_forked_contexts_share_tables_and_prove_the_same_bytes(cvname, gates):
    """zkt_ctx_fork: contexts that share one context's SRS / Lagrange / circuit / twiddle tables and own only their work
    buffers.  The parent and two forks prove different proofs concurrently (one thread each): the oracle's bytes everywhere.
    Lifetime rules: a parent with live forks refuses to reload its key or circuit; destroyed first it lingers until its last
    fork is gone (the forks keep proving); a fork that loads its own key stops sharing and still proves the same bytes."""
    import threading
    import zkt_plonk_amd as z
    cv = F.CURVES[cvname]
    cs = P.synthetic_circuit(cv, gates, 64, seed=gates + 5)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0xF04C + gates, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    pkm = {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64) for k in z.PK_ORDER}
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    pi_pos = sorted(cs.pi)
    pi_vals = K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos])
    table = K.fr_to_mont(cv, cs.table)
    bl = [field_elems(cv.fr.p, 7700 + t, P.NUM_BLINDERS) for t in range(3)]
    want = [P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), x).serialize(cv) for x in bl]

    def tr():
        return z.seed_transcript(z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8), vk.n, vk.commits)

    parent = z.Context(cv.name, 0)
    parent.srs_load(srs_arr)
    z.GpuProver(parent, n.bit_length() - 1, pkm)
    assert parent.prove(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, bl[0]), tr()) == want[0]   # builds the Lagrange table
    forks = [parent.fork(), parent.fork()]
    ctxs3 = [parent] + forks
    for f in forks:
        assert f.lagrange_info() == parent.lagrange_info() and f.msm_info() == parent.msm_info()
    with pytest.raises(z.ZktError):
        parent.srs_load(srs_arr)                  # its tables are in use
    with pytest.raises(z.ZktError):
        z.GpuProver(parent, n.bit_length() - 1, pkm)
    errors = []

    def worker(t):
        try:
            for _ in range(3):
                got = ctxs3[t].prove(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, bl[t]), tr())
                assert got == want[t], t
        except BaseException as e:
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=600)
    assert not errors, errors
    parent.close()                                # lingers: the forks still read its tables
    assert forks[0].prove(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, bl[1]), tr()) == want[1]
    forks[1].srs_load(srs_arr)                    # a key of its own: stops sharing that part, same bytes
    assert forks[1].prove(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, bl[2]), tr()) == want[2]
    forks[0].close()
    assert forks[1].prove(a, b, c, table, pi_pos, pi_vals, K.fr_to_mont(cv, bl[0]), tr()) == want[0]
    forks[1].close()                              # the last fork: the parent goes with it
