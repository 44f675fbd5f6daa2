"""Pins the C++ part of the oracle against the Python big-integer oracle (CPU)."""
import numpy as np
import pytest

from oracle import fields as F, curve as C, plonk as P, coracle as K
from oracle.ntt import Domain
from helpers import field_elems, digest, unhex_point


def test_parameter_tables():
    for i, f in enumerate((F.BN254_FR, F.BN254_FQ, F.BLS12_381_FR, F.BLS12_381_FQ)):
        assert K.params(i) == dict(p=f.p, inv=f.inv64, r=f.R, r2=f.R2)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_ntt_matches_python_and_golden(cv, golden):
    p = cv.fr.p
    for e in golden[cv.name]["ntt"]:
        log_n = e["n"].bit_length() - 1
        x = K.fr_to_mont(cv, field_elems(p, e["seed"], e["in_len"]))
        for k, inv, cos in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
            y = K.fr_from_mont(cv, K.ntt_mont(cv, log_n, inv, cos, x))
            assert digest(y) == e[k + "_sha256"]
    for log_n in (0, 1, 2, 5):
        d = Domain(cv.fr, 1 << log_n)
        x = field_elems(p, 4242 + log_n, 1 << log_n)
        assert K.fr_from_mont(cv, K.ntt_mont(cv, log_n, 0, 1, K.fr_to_mont(cv, x))) == d.coset_fft(x)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_srs_and_msm_match_python_and_golden(cv, golden):
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    p = cv.fr.p
    sm = K.srs_mont(cv, tau, 1000)
    assert K.points_from_mont(cv, sm[:4]) == [unhex_point(q) for q in g["srs_first"]]
    for e in g["msm"]:
        n = e["n"]
        if e["seed"] is None:
            sc = [int(s, 16) for s in e["scalars"]]
        else:
            sc = field_elems(p, e["seed"], n)
            if n >= 31:
                sc[0], sc[1], sc[2], sc[5] = 0, 1, p - 1, 0
        out, inf = K.msm_mont(cv, sm[:n], K.fr_to_mont(cv, sc))
        got = None if inf else K.points_from_mont(cv, out)[0]
        assert got == unhex_point(e["result"])
        # canonical (non-Montgomery) scalar entry point
        out2, inf2 = K.msm_mont(cv, sm[:n], K.ints_to_limbs(sc, 4), scalars_mont=False)
        assert (None if inf2 else K.points_from_mont(cv, out2)[0]) == got


def test_cbackend_proof_equals_python_backend(golden):
    cv = F.BN254
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    be = K.CBackend(cv, K.srs_mont(cv, tau, 4 * n + 1))
    pk, epk, vk = P.setup(be, None_srs(4 * n + 1), cs, True)
    proof = P.prove(be, None_srs(4 * n + 1), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk),
                    field_elems(cv.fr.p, g["test_circuit"]["blinder_seed"], P.NUM_BLINDERS))
    assert proof.serialize(cv).hex() == g["test_circuit"]["proof_bytes"]


def None_srs(count):
    return [None] * count  # CBackend.msm takes its bases from srs_arr; only the length is checked


# ---- the array prover (oracle/fastplonk.py + the C++ loops) against the big-integer prover (oracle/plonk.py) ----
from oracle import fastplonk as FP
from oracle.ntt import trim, poly_eval


def _mont(cv, vals):
    return K.fr_to_mont(cv, vals) if len(vals) else np.zeros((0, 4), dtype=np.uint64)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_prover_loops_match_python(cv):
    """z1 / z2 grand products, combine_split, quotient evaluations, scaled sums, evaluation and the opening quotient:
    each C++ loop against its twin in oracle/plonk.py on a 256-row circuit."""
    p = cv.fr.p
    cs = P.synthetic_circuit(cv, 200, 16, seed=3)
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    be = K.CBackend(cv, K.srs_mont(cv, 99, n + 8))
    pk, epk, vk = P.setup(be, None_srs(n + 8), cs, True)
    dom = be.domain(n)
    a, b, c = cs.wire_evals(n)
    beta, gamma, delta, eps, alpha = field_elems(p, 17, 5)
    want = P.compute_z1_evals(cv, dom, beta, gamma, a, b, c, epk.sigma1, epk.sigma2, epk.sigma3)
    got = K.z1_evals(cv, log_n, _mont(cv, [beta]), _mont(cv, [gamma]), _mont(cv, a), _mont(cv, b), _mont(cv, c),
                     _mont(cv, epk.sigma1), _mont(cv, epk.sigma2), _mont(cv, epk.sigma3))
    assert K.fr_from_mont(cv, got) == want
    assert K.fr_from_mont(cv, K.domain_points(cv, log_n)) == dom.elements()
    t_ev = list(cs.table) + [0] * (n - len(cs.table))
    f_ev = [q * x % p for q, x in zip(epk.q_lookup, c)]
    assert K.fr_from_mont(cv, K.vec_op(cv, "mul", _mont(cv, epk.q_lookup), _mont(cv, c))) == f_ev
    assert K.fr_from_mont(cv, K.vec_op(cv, "sub", _mont(cv, a), _mont(cv, b))) == [(x - y) % p for x, y in zip(a, b)]
    h1, h2 = P.combine_split(t_ev, f_ev)
    g1, g2 = K.combine_split(_mont(cv, t_ev), _mont(cv, f_ev))
    assert K.fr_from_mont(cv, g1) == h1 and K.fr_from_mont(cv, g2) == h2
    with pytest.raises(KeyError):
        K.combine_split(_mont(cv, t_ev), _mont(cv, [f_ev[0], (max(t_ev) + 1) % p]))
    want = P.compute_z2_evals(cv, dom, delta, eps, f_ev, t_ev, h1, h2)
    got = K.z2_evals(cv, log_n, _mont(cv, [delta]), _mont(cv, [eps]), _mont(cv, f_ev), _mont(cv, t_ev), g1, g2)
    assert K.fr_from_mont(cv, got) == want
    # quotient evaluations on random "witness cosets" (the identity does not need to hold for the loop to be compared)
    wit = {k: field_elems(p, 100 + i, 4 * n) for i, k in enumerate(K.WIT_ORDER)}
    want = P.quotient_evals(cv, n, epk, (alpha, beta, gamma, delta, eps), wit)
    got = K.quotient_evals(cv, log_n, _mont(cv, [alpha, beta, gamma, delta, eps]),
                           {k: _mont(cv, v) for k, v in epk.cosets.items()}, {k: _mont(cv, v) for k, v in wit.items()})
    assert K.fr_from_mont(cv, got) == want
    # scaled sums, evaluation, division by (X - z)
    polys = [field_elems(p, 7, 50), field_elems(p, 8, 77), [], field_elems(p, 9, 3)]
    sc = field_elems(p, 10, 4)
    want = P.poly_add(p, *[P.poly_scale(p, q, s) for q, s in zip(polys, sc)])
    got = K.lincomb(cv, [_mont(cv, q) for q in polys], _mont(cv, sc), 80)
    assert K.fr_from_mont(cv, got) == want + [0] * (80 - len(want))
    big = field_elems(p, 11, 20000)
    z = field_elems(p, 12, 1)[0]
    assert K.fr_from_mont(cv, K.poly_eval(cv, _mont(cv, big), _mont(cv, [z])).reshape(1, 4)) == [poly_eval(cv.fr, big, z)]
    q = [0] * (len(big) - 1)
    carry = 0
    for i in range(len(big) - 1, 0, -1):
        carry = (big[i] + z * carry) % p
        q[i - 1] = carry
    assert K.fr_from_mont(cv, K.div_linear(cv, _mont(cv, big), _mont(cv, [z]))) == q
    assert K.trim_len(_mont(cv, [1, 0, 5, 0, 0])) == 3 and K.trim_len(_mont(cv, [0, 0])) == 0


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("gates,table_size,n_public", [(60, 16, 2), (1000, 64, 7), (4000, 1024, 0)])
def test_array_prover_equals_python_prover(cv, gates, table_size, n_public):
    """Whole proofs: oracle.fastplonk (arrays + C++ loops) == oracle.plonk (Python integers), byte for byte, and the
    setup commitments agree; the restated verifier accepts."""
    cs = P.synthetic_circuit(cv, gates, table_size, seed=gates, n_public=n_public)
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    tau = 0xFEED + gates
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, None_srs(n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 5 + gates, P.NUM_BLINDERS)
    want = P.prove(be, None_srs(n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders)
    evals = {k: _mont(cv, v) for k, v in P.setup_evals(be, cs).items()}
    keys = FP.setup(cv, srs, log_n, evals)
    assert keys.commits == vk.commits
    for k in P.PK_POLYS:
        assert K.fr_from_mont(cv, keys.pk[k]) == pk.polys[k]
    for k in K.EPK_ORDER:
        assert K.fr_from_mont(cv, keys.epk[k]) == epk.cosets[k], k
    a, b, c = cs.wire_evals(cs.n_gates)
    tr = P.new_seeded_transcript(cv, keys.verifier_key(cv, cs.pi.keys()))
    got = FP.prove(cv, srs, keys, _mont(cv, a), _mont(cv, b), _mont(cv, c), _mont(cv, cs.table), cs.pi, tr, blinders)
    assert got == want.serialize(cv)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), pis)


def test_array_prover_reference_test_circuit_golden(golden):
    """The committed golden proof of the reference's TestCircuit (plonk.rs:144-218) through the array prover."""
    for cv in (F.BN254, F.BLS12_381):
        g = golden[cv.name]
        cs = P.test_circuit(cv)
        n = cs.circuit_bound()
        srs = K.srs_mont(cv, int(g["tau"], 16), n + 8)
        be = K.CBackend(cv, srs)
        keys = FP.setup(cv, srs, n.bit_length() - 1, {k: _mont(cv, v) for k, v in P.setup_evals(be, cs).items()})
        a, b, c = cs.wire_evals(cs.n_gates)
        tr = P.new_seeded_transcript(cv, keys.verifier_key(cv, cs.pi.keys()))
        got = FP.prove(cv, srs, keys, _mont(cv, a), _mont(cv, b), _mont(cv, c), _mont(cv, cs.table), cs.pi, tr,
                       field_elems(cv.fr.p, g["test_circuit"]["blinder_seed"], P.NUM_BLINDERS))
        assert got.hex() == g["test_circuit"]["proof_bytes"]


def test_config0_bn254_2_14_cpu_proof_verifies():
    """BASELINE.json configs[0] (BN254, domain 2^14, CPU path, no GPU): the bench workload's shape at 2^14 proved by the
    CPU oracle's array prover and accepted by the restated verifier; a tampered proof is rejected."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    cv = F.BN254
    log_n, n = 14, 1 << 14
    tau = 0x5EED5EED1234567890ABCDEF % cv.fr.p
    circ = B.synthetic_circuit(B.FIELDS[cv.name], log_n)
    srs = K.srs_mont(cv, tau, n + 8)
    keys = FP.setup(cv, srs, log_n, {k: _mont(cv, circ["sel"][k]) for k in P.PK_POLYS})
    vk = keys.verifier_key(cv, circ["pi"].keys())
    g = circ["gates"]
    proof = FP.prove(cv, srs, keys, _mont(cv, circ["a"][:g]), _mont(cv, circ["b"][:g]), _mont(cv, circ["c"][:g]),
                     _mont(cv, circ["table"]), circ["pi"], P.new_seeded_transcript(cv, vk), field_elems(cv.fr.p, 14, P.NUM_BLINDERS))
    assert len(proof) == 802
    pis = [circ["pi"][k] for k in sorted(circ["pi"])]
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, proof), P.new_seeded_transcript(cv, vk), pis)
    bad = bytearray(proof)
    bad[-70] ^= 4
    assert not P.verify(cv, tau, vk, P.proof_deserialize(cv, bytes(bad)), P.new_seeded_transcript(cv, vk), pis)
