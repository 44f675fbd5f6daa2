"""f4: the verifier without its pairings (zkt_verify_prepare, host-only C++) against the oracle's
restatement of proof_system/proof.rs:285-503 -- same (L, W) pairs, and the trapdoor form of the pairing check
(L == tau W) accepts honest proofs and rejects tampered ones.  Merlin and Ethereum transcripts, both curves."""
import numpy as np
import pytest

from oracle import fields as F, plonk as P, coracle as K, curve as C
from helpers import field_elems
import zkt_plonk_amd as z
from zkt_plonk_amd import _lib


def _proof(cv, kind, gates=150, n_public=3, seed=5):
    cs = P.synthetic_circuit(cv, gates, 16, seed=seed, n_public=n_public)
    n = cs.circuit_bound()
    tau = 0x7E57ED + seed
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    proof = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk, kind),
                    field_elems(cv.fr.p, 70 + seed, P.NUM_BLINDERS))
    return cs, tau, srs, vk, proof


def _prepare(cv, kind, vk, srs, pis, proof_bytes):
    tr = z.Transcript(kind, "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
    z.seed_transcript(tr, vk.n, vk.commits)
    names = z.PK_ORDER
    commits = K.points_to_mont(cv, [vk.commits[k] for k in names])
    inf = [vk.commits[k] is None for k in names]
    return _lib.verify_prepare(cv.name, vk.n, commits, inf, K.fr_to_mont(cv, vk.pi_roots) if pis else np.zeros((0, 4), np.uint64),
                               K.fr_to_mont(cv, pis) if pis else np.zeros((0, 4), np.uint64), proof_bytes, srs[0], tr)


@pytest.mark.parametrize("cv,kind", [(F.BN254, "merlin"), (F.BLS12_381, "merlin"), (F.BN254, "ethereum")],
                         ids=["bn254-merlin", "bls12_381-merlin", "bn254-ethereum"])
def test_verify_prepare_matches_the_oracle_and_the_trapdoor_check(cv, kind):
    cs, tau, srs, vk, proof = _proof(cv, kind)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    raw = proof.serialize(cv)
    want = P.verify_prepare(cv, vk, P.proof_deserialize(cv, raw), P.new_seeded_transcript(cv, vk, kind), pis)
    got, inf = _prepare(cv, kind, vk, srs, pis, raw)
    pts = [None if inf[i] else K.points_from_mont(cv, got[i:i + 1])[0] for i in range(4)]
    assert [(pts[0], pts[1]), (pts[2], pts[3])] == want
    for L, W in ((pts[0], pts[1]), (pts[2], pts[3])):
        assert L == C.scalar_mul(cv, tau, W)                        # e(L, h) == e(W, tau h)
    # a flipped evaluation, a swapped commitment and a wrong public input are all rejected by the same identity
    bad = bytearray(raw)
    bad[-40] ^= 1
    nb = (cv.fq.bits + 2 + 7) // 8
    swapped = raw[nb:2 * nb] + raw[:nb] + raw[2 * nb:]
    for tampered, tp in ((bytes(bad), pis), (swapped, pis), (raw, [(pis[0] + 1) % cv.fr.p] + pis[1:])):
        g2, i2 = _prepare(cv, kind, vk, srs, tp, tampered)
        p2 = [None if i2[i] else K.points_from_mont(cv, g2[i:i + 1])[0] for i in range(4)]
        assert not (p2[0] == C.scalar_mul(cv, tau, p2[1]) and p2[2] == C.scalar_mul(cv, tau, p2[3]))


def test_verify_prepare_rejects_malformed_proofs():
    cv = F.BN254
    cs, tau, srs, vk, proof = _proof(cv, "merlin", seed=9)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    raw = proof.serialize(cv)
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, raw[:-1])                          # wrong length
    off = bytearray(raw)
    x = int.from_bytes(raw[:31], "little")
    for delta in range(1, 50):                                                  # an x with no point above it
        cand = (x + delta) % cv.fq.p
        if pow((cand ** 3 + cv.b) % cv.fq.p, (cv.fq.p - 1) // 2, cv.fq.p) != 1:
            off[:32] = cand.to_bytes(32, "little")
            break
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, bytes(off))
    nonc = bytearray(raw)
    nonc[-32:] = (cv.fr.p).to_bytes(32, "little")                               # evaluation = modulus: not canonical
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, bytes(nonc))
    some = bytearray(raw)
    some[11 * 32 + 32] = 1                                                      # kzg10::Proof::random_v must be None
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, bytes(some))


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_multi_scalar_mul_on_arbitrary_points_matches_the_oracle(cv):
    """HomomorphicCommitment::multi_scalar_mul (commitment.rs:32-45): 13 arbitrary points (one of them the identity),
    Montgomery and canonical scalars, zero / one / r - 1 among them; and the empty sum."""
    p = cv.fr.p
    srs = K.srs_mont(cv, 0xABCD, 13)
    pts = K.points_from_mont(cv, srs)
    pts[4] = None
    srs[4] = 0
    sc = field_elems(p, 77, 13)
    sc[0], sc[1], sc[2] = 0, 1, p - 1
    want = C.msm_naive(cv, pts, sc)
    got, inf = _lib.g1_msm_host(cv.name, srs, K.fr_to_mont(cv, sc))
    assert (None if inf else K.points_from_mont(cv, got.reshape(1, -1))[0]) == want
    got2, inf2 = _lib.g1_msm_host(cv.name, srs, K.ints_to_limbs(sc, 4), montgomery=False)
    assert inf2 == inf and np.array_equal(got2, got)
    L = cv.fq.limbs64
    assert _lib.g1_msm_host(cv.name, np.zeros((0, 2 * L), np.uint64), np.zeros((0, 4), np.uint64))[1]


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_host_pairing_shortcuts_agree_with_their_definitions(cv):
    """csrc/pairing.hpp selftest: Frobenius maps = powers by p, sparse / complex / cyclotomic products = the dense ones,
    exponentiation by x on cyclotomic squarings = plain square-and-multiply, the final exponentiation leaves order r."""
    assert z.lib().zkt_debug_pairing_selftest(0 if cv.name == "bn254" else 1) == 0


def _point_outside_g1(cv):
    """A point of E(Fq) of order dividing the cofactor (BLS12-381: h = (x - 1)^2 / 3 > 1): [r] of a random curve point."""
    p = cv.fq.p
    x = 5
    while True:
        y = C.sqrt_mod((x ** 3 + cv.b) % p, p)
        if y is not None:
            small = C.add(cv, C.scalar_mul(cv, cv.fr.p - 1, (x, y)), (x, y))    # [r] P (scalar_mul reduces modulo r)
            if small is not None:
                return small
        x += 1


def test_checked_deserialisation_rejects_points_outside_the_prime_order_subgroup():
    """proof.rs:308: "subgroup checks are done when the proof is deserialised" (ark-serialize's checked path).  BLS12-381
    has cofactor > 1: a commitment shifted by a point of cofactor order is on the curve and must be refused; so is such
    a point handed to the pairing check.  (BN254 has cofactor one: every curve point is in G1.)"""
    cv = F.BLS12_381
    cs, tau, srs, vk, proof = _proof(cv, "merlin", seed=11)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    raw = proof.serialize(cv)
    _prepare(cv, "merlin", vk, srs, pis, raw)                                   # the honest proof passes
    small = _point_outside_g1(cv)
    assert C.is_on_curve(cv, small) and C.add(cv, C.scalar_mul(cv, cv.fr.p - 1, small), small) is not None
    nb = 48
    for k in (0, 6, 11):                                                        # a_commit, z1_commit, the first opening
        off = k * nb + (1 if k > 11 else 0)
        honest = C.point_deserialize_compressed(cv, raw[off:off + nb])
        shifted = C.add(cv, honest, small)
        assert C.is_on_curve(cv, shifted)
        bad = raw[:off] + C.point_serialize_compressed(cv, shifted) + raw[off + nb:]
        with pytest.raises(z.ZktError):
            _prepare(cv, "merlin", vk, srs, pis, bad)
    # the small-order point itself, and the pairing entry point
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, C.point_serialize_compressed(cv, small) + raw[nb:])
    from oracle import pairing as PR
    from test_pairing_host import g2_mont
    H = PR.G2_GENERATORS[cv.name]
    with pytest.raises(_lib.ZktError):
        _lib.pairing_product_is_one(cv.name, K.points_to_mont(cv, [small]), g2_mont(cv, [H]))
    # every multiple of the generator passes the endomorphism test (it is the subgroup check, not a filter on x)
    G = C.generator(cv)
    for k in (1, 2, 3, 0xd201000000010000, cv.fr.p - 1):
        assert _lib.pairing_product_is_one(cv.name, K.points_to_mont(cv, [C.scalar_mul(cv, k, G), None]), g2_mont(cv, [None, H]))


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_malformed_flag_bytes_are_refused(cv):
    """ark-serialize SWFlags::from_u8: both flag bits at once is no encoding.  Under the infinity flag GroupAffine::deserialize
    (ark-ec 0.3, as recalled: "parity unpinned") parses x like any field element and then returns zero(): a canonical x is
    ignored, an x at or above the modulus is an error."""
    cs, tau, srs, vk, proof = _proof(cv, "merlin", seed=13)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    raw = bytearray(proof.serialize(cv))
    nb = cv.fq.limbs64 * 8
    both = bytearray(raw)
    both[nb - 1] |= 0xC0
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, bytes(both))
    inf_with_x = bytearray(raw)
    inf_with_x[nb - 1] = (inf_with_x[nb - 1] & 0x3F) | 0x40                    # infinity flag over a non-zero x: ignored
    clean_inf = bytearray(raw)
    clean_inf[:nb] = bytes(nb - 1) + b"\x40"                                    # the usual encoding of the identity
    got_x, inf_x = _prepare(cv, "merlin", vk, srs, pis, bytes(inf_with_x))
    got_c, inf_c = _prepare(cv, "merlin", vk, srs, pis, bytes(clean_inf))
    assert np.array_equal(got_x, got_c) and list(inf_x) == list(inf_c)
    big_x = bytearray(raw)
    big_x[:nb] = cv.fq.p.to_bytes(nb, "little")                                 # x = p under the infinity flag
    big_x[nb - 1] |= 0x40
    with pytest.raises(z.ZktError):
        _prepare(cv, "merlin", vk, srs, pis, bytes(big_x))
