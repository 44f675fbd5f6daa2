"""Reference property tests and KATs restated against the oracle prover (CPU)."""
import pytest

from oracle import fields as F, curve as C, plonk as P
from oracle.ntt import Domain, poly_eval, trim
from helpers import field_elems, unhex_point


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_combine_split_reference_kat(cv):
    # plonk-core/src/lookup/multiset.rs:272-329
    t = [0, 1, 2, 3, 4, 5, 6]
    f = [3, 6, 0, 5, 4, 3, 2, 0, 0, 1, 2]
    h1, h2 = P.combine_split(t, f)
    assert h1 == [0, 0, 1, 2, 2, 3, 4, 5, 6]
    assert h2 == [0, 0, 1, 2, 3, 3, 4, 5, 6]
    with pytest.raises(KeyError):
        P.combine_split(t, [7])


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_add_blinders_keeps_domain_evals(cv):
    # plonk-core/src/proof_system/prove.rs:498-526
    f = cv.fr
    d = Domain(f, 8)
    evals = field_elems(f.p, 11, 8)
    poly = trim(d.ifft(evals))
    for k in (1, 2, 3):
        blinded = P.add_blinders_to_poly(f.p, poly, field_elems(f.p, 100 + k, k))
        assert len(blinded) == 8 + k
        assert [poly_eval(f, blinded, w) for w in d.elements()] == evals


def test_z2_grand_product_identity():
    # plonk-core/src/lookup/mod.rs:170-233 (Bn254)
    cv = F.BN254
    f = cv.fr
    p = f.p
    t = [0, 0, 1, 2, 3, 4, 5, 6]
    fq = [3, 6, 0, 5, 4, 3, 2, 0]
    h1, h2 = P.combine_split(t, fq)
    d = Domain(f, 8)
    delta, epsilon = field_elems(p, 5, 2)
    z2 = P.compute_z2_evals(cv, d, delta, epsilon, fq, t, h1, h2)
    assert z2[0] == 1
    opd = (1 + delta) % p
    e = epsilon * opd % p
    for i in range(8):
        j = (i + 1) % 8
        lhs = opd * (epsilon + fq[i]) % p * (delta * t[j] + e + t[i]) % p * z2[i] % p
        rhs = (delta * h2[i] + e + h1[i]) * (delta * h1[j] + e + h2[i]) % p * z2[j] % p
        assert lhs == rhs


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_z1_grand_product_identity(cv):
    # plonk-core/src/permutation/mod.rs:328-392 on a small copy-constrained circuit
    cs = P.ConstraintSystem(cv, [1], 4)
    x1, x2, x3, x4 = (cs.assign_variable(v) for v in (4, 12, 8, 3))
    cs.arith_constrain(x1, x4, x2, q_m=1, q_o=-1)       # 4*3 = 12
    cs.arith_constrain(x1, x3, x2, q_l=1, q_r=1, q_o=-1)  # 4+8 = 12
    cs.arith_constrain(x3, x1, x2, q_l=1, q_r=1, q_o=-1)
    cs.arith_constrain(x4, x1, x2, q_m=1, q_o=-1)
    assert cs.check_satisfied()
    n = 8
    p = cv.fr.p
    d = Domain(cv.fr, n)
    roots = d.elements()
    sig = cs.sigma_mappings(n)
    ks = (1, F.K1, F.K2)
    s = [[ks[c] * roots[i] % p for (c, i) in sig[j]] for j in range(3)]
    a, b, c = cs.wire_evals(n)
    beta, gamma = field_elems(p, 9, 2)
    z1 = P.compute_z1_evals(cv, d, beta, gamma, a, b, c, *s)
    assert z1[0] == 1
    for i in range(n):
        j = (i + 1) % n
        num = (beta * roots[i] + a[i] + gamma) * (F.K1 * beta * roots[i] + b[i] + gamma) % p \
            * (F.K2 * beta * roots[i] + c[i] + gamma) % p
        den = (beta * s[0][i] + a[i] + gamma) * (beta * s[1][i] + b[i] + gamma) % p * (beta * s[2][i] + c[i] + gamma) % p
        assert z1[i] * num % p == z1[j] * den % p


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_full_prove_verify_and_golden_bytes(cv, golden):
    # plonk-core/src/plonk.rs:191-218 test_full (KZG) + the committed proof bytes
    g = golden[cv.name]["test_circuit"]
    tau = int(golden[cv.name]["tau"], 16)
    cs = P.test_circuit(cv)
    assert cs.check_satisfied()
    n = cs.circuit_bound()
    assert n == g["n"] == 128
    be = P.Backend(cv)
    srs = C.srs_powers(cv, tau, n + 8)
    pk, epk, vk = P.setup(be, srs, cs, True)
    assert {k: unhex_point(v) for k, v in g["vk"].items()} == vk.commits
    blinders = field_elems(cv.fr.p, g["blinder_seed"], P.NUM_BLINDERS)
    tr = P.ProverTrace()
    proof = P.prove(be, srs, pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders, tr)
    data = proof.serialize(cv)
    assert len(data) == (802 if cv.name == "bn254" else 1010)
    assert data.hex() == g["proof_bytes"]
    assert {k: "%x" % v for k, v in tr.challenges.items()} == g["challenges"]
    assert P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [10, 2])
    assert not P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [10, 3])
    # the wire format parses back to the same proof (decompression picks y by the sign flag)
    back = P.proof_deserialize(cv, data)
    assert back.commits == proof.commits and back.aw_opening == proof.aw_opening
    assert back.saw_opening == proof.saw_opening and back.evaluations == proof.evaluations
    assert back.serialize(cv) == data
    # tampering with an evaluation must be rejected
    proof.evaluations.a = (proof.evaluations.a + 1) % cv.fr.p
    assert not P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [10, 2])


def test_synthetic_circuit_is_satisfied_and_proves():
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 60, 16, seed=3)
    assert cs.check_satisfied() and cs.circuit_bound() == 64
    tau = 987654321
    srs = C.srs_powers(cv, tau, 64 + 8)
    be = P.Backend(cv)
    pk, epk, vk = P.setup(be, srs, cs, True)
    proof = P.prove(be, srs, pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), field_elems(cv.fr.p, 1, P.NUM_BLINDERS))
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    assert P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), pis)
