"""Pins the oracle's primitives: published constants, the reference's literal KATs, and
definition-level cross-checks (naive DFT, double-and-add)."""
import hashlib

import pytest

from oracle import fields as F, curve as C, transcript as T
from oracle.ntt import Domain, dft_naive
from helpers import field_elems, digest, unhex_point


def test_published_field_constants():
    # SURVEY.md section 8c (ark-bn254 / ark-bls12-381 0.3 FftParameters / FpParameters)
    assert F.BN254_FR.two_adic_root == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert F.BLS12_381_FR.two_adic_root == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert F.BN254_FR.R == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert F.BN254_FR.inv64 == 0xc2e1f593efffffff
    assert F.BLS12_381_FR.R == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    assert F.BLS12_381_FR.inv64 == 0xfffffffeffffffff
    for f in (F.BN254_FR, F.BLS12_381_FR):
        w = f.two_adic_root
        assert pow(w, 1 << f.two_adicity, f.p) == 1 and pow(w, 1 << (f.two_adicity - 1), f.p) == f.p - 1


@pytest.mark.parametrize("f", [F.BN254_FR, F.BLS12_381_FR])
def test_k1_k2_cosets(f):
    # plonk-core/src/permutation/constants.rs:36-50 test_constants
    n = 1 << f.two_adicity
    assert pow(F.K1, n, f.p) != 1
    assert pow(F.K1 * f.inv(F.K2), n, f.p) != 1


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_curve_generators(cv):
    G = C.generator(cv)
    assert C.is_on_curve(cv, G)
    assert C.scalar_mul(cv, cv.fr.p, G) is None or C.scalar_mul(cv, cv.fr.p - 1, G) == C.neg(cv, G)
    assert C.add(cv, C.scalar_mul(cv, cv.fr.p - 1, G), G) is None
    assert C.scalar_mul(cv, 5, G) == C.add(cv, C.double(cv, C.double(cv, G)), G)


def test_keccak_against_hashlib():
    for m in (b"", b"abc", bytes(range(256)) * 3):
        assert T.sha3_256(m) == hashlib.sha3_256(m).digest()


# Public known answers that come from neither the reference nor this repository: legacy Keccak-256 of "" and "abc"
# (the Ethereum yellow paper's hash; hashlib has no legacy padding) and G + G on alt_bn128 (EIP-196's bn256Add vector).
KECCAK_EMPTY = "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
KECCAK_ABC = "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
BN254_2G = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
            0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)


def test_public_known_answers_keccak_and_bn254_doubling():
    assert T.keccak256(b"").hex() == KECCAK_EMPTY
    assert T.keccak256(b"abc").hex() == KECCAK_ABC
    G = C.generator(F.BN254)
    assert C.add(F.BN254, G, G) == BN254_2G == C.scalar_mul(F.BN254, 2, G) == C.double(F.BN254, G)
    # ... and the product's own host arithmetic (zkt_g1_sum_host, zkt_g1_msm_host, the Keccak transcript's hash)
    import numpy as np
    from oracle import coracle as K
    from zkt_plonk_amd import _lib
    g = K.points_to_mont(F.BN254, [G, G])
    out, inf = _lib.g1_sum_host("bn254", g)
    assert not inf and K.points_from_mont(F.BN254, out.reshape(1, -1))[0] == BN254_2G
    out, inf = _lib.g1_msm_host("bn254", g[:1], K.fr_to_mont(F.BN254, [2]))
    assert not inf and K.points_from_mont(F.BN254, np.asarray(out).reshape(1, -1))[0] == BN254_2G


def test_merlin_conformance_vector():
    # merlin's published cross-implementation vector ("test protocol" / "some label" / "some data")
    t = T.Merlin(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == \
        "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_ethereum_transcript_reference_kat():
    # gadgets/src/transcript.rs:101-127 (the reference's only byte-level KAT)
    e = T.EthereumTranscript(F.BN254, "test")
    e.append_u64("a", 1)
    assert e.challenge_scalar("a").to_bytes(32, "big").hex() == \
        "0f9d11cec4f06b0d18060cde3db4196495ddfbb096108951446fc8a1d45f4b59"
    e.append_scalar("b", 2)
    assert e.challenge_scalar("b").to_bytes(32, "big").hex() == \
        "0f4dccb919a5dba2dd010a562ba45b4551291f5e565706536e78b24ac8b5c64d"
    e.append_commitment("c", (3, 4))
    assert e.challenge_scalar("c").to_bytes(32, "big").hex() == \
        "1b5bf46adfcd1dd4f9ac7166586cf83f261192bc4b83fdda30ddee22f9054c1f"


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_ntt_against_definition_and_golden(cv, golden):
    f = cv.fr
    for n in (1, 2, 8, 32):
        d = Domain(f, n)
        x = field_elems(f.p, 77 + n, n)
        assert d.fft(x) == dft_naive(f, x, n, d.group_gen)
        assert d.ifft(d.fft(x)) == x
        g = f.generator
        assert d.coset_fft(x) == dft_naive(f, [c * pow(g, i, f.p) % f.p for i, c in enumerate(x)], n, d.group_gen)
        assert d.coset_ifft(d.coset_fft(x)) == x
    for e in golden[cv.name]["ntt"]:
        d = Domain(f, e["n"])
        x = field_elems(f.p, e["seed"], e["in_len"])
        for k, fn in (("fft", d.fft), ("ifft", d.ifft), ("coset_fft", d.coset_fft), ("coset_ifft", d.coset_ifft)):
            y = fn(x)
            assert digest(y) == e[k + "_sha256"]
            if k in e:
                assert [int(v, 16) for v in e[k]] == y


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_msm_golden_small(cv, golden):
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    srs = C.srs_powers(cv, tau, 33)
    assert [unhex_point(p) for p in g["srs_first"]] == srs[:4]
    for e in g["msm"]:
        if e["n"] > 33:
            continue
        sc = [int(s, 16) for s in e["scalars"]]
        assert C.msm_pippenger(cv, srs[:e["n"]], sc) == unhex_point(e["result"])
        assert C.msm_naive(cv, srs[:e["n"]], sc) == unhex_point(e["result"])


def test_msm_window_rule():
    # c = 11/14/15/15/17 for n = 2^14/18/19/20/22; 3 below 32
    assert [C.msm_window_bits(1 << k) for k in (14, 18, 19, 20, 22)] == [11, 14, 15, 15, 17]
    assert C.msm_window_bits(31) == 3 and C.msm_window_bits(32) == 5
    assert C.msm_reference_adds(1 << 20, 254) == 17 * (1 << 20) + 17 * 2 * 32767


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_point_serialization_roundtrip(cv):
    G = C.generator(cv)
    for k in (1, 2, 3, 12345, cv.fr.p - 1):
        P_ = C.scalar_mul(cv, k, G)
        b = C.point_serialize_compressed(cv, P_)
        assert len(b) == (32 if cv.name == "bn254" else 48)
        assert C.point_deserialize_compressed(cv, b) == P_
    z = C.point_serialize_compressed(cv, None)
    assert z[-1] == 0x40 and C.point_deserialize_compressed(cv, z) is None
    assert len(C.point_to_bytes_uncompressed(cv, G)) == (65 if cv.name == "bn254" else 97)


@pytest.mark.parametrize("f", [F.BN254_FR, F.BLS12_381_FR], ids=["bn254", "bls12_381"])
def test_ntt_against_sympy(f):
    """An independent implementation that is neither the reference's nor this repository's: sympy.discrete.transforms
    ntt / intt (Cooley-Tukey over Z_p with the root g^((p-1)/n), g = sympy's smallest primitive root -- 5 for the BN254
    scalar field, 7 for BLS12-381's: arkworks' GENERATOR).  oracle/ntt.py's fft / ifft and, through the
    definition c_j g^j, coset_fft / coset_ifft must give the same numbers for n = 8, 64, 1024 (natural order both ways,
    inverse scaled by 1/n)."""
    from sympy import primitive_root
    from sympy.discrete.transforms import ntt, intt
    p = f.p
    assert primitive_root(p) == f.generator
    for n in (8, 64, 1024):
        dom = Domain(f, n)
        coeffs = field_elems(p, 4242 + n, n)
        ev = dom.fft(coeffs)
        assert ev == [int(x) for x in ntt(coeffs, prime=p)]
        assert dom.ifft(ev) == coeffs == [int(x) for x in intt(ev, prime=p)]
        short = coeffs[:n // 2 + 1]                       # zero-padded input, as the prover feeds it
        assert dom.fft(short) == [int(x) for x in ntt(short + [0] * (n - len(short)), prime=p)]
        g = f.generator
        shifted = [c * pow(g, j, p) % p for j, c in enumerate(coeffs)]
        cev = dom.coset_fft(coeffs)
        assert cev == [int(x) for x in ntt(shifted, prime=p)]
        assert dom.coset_ifft(cev) == coeffs
        assert dom.coset_ifft(ev) == [int(x) * pow(g, -j, p) % p for j, x in enumerate(intt(ev, prime=p))]


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_curve_arithmetic_and_msm_against_sympy(cv):
    """A second independent implementation: sympy.ntheory.elliptic_curve.EllipticCurve (affine chord-and-tangent over Z_q)
    on y^2 = x^3 + b with the published G1 generators.  Scalar multiples, sums, doublings, the order of the generator, a
    small multi-scalar multiplication (oracle.curve's Pippenger and the C++ port's) and a short SRS (powers of a trapdoor)
    must agree -- the group law and the MSM results rest on more than this repository's own code."""
    from sympy.ntheory.elliptic_curve import EllipticCurve
    from oracle import coracle as K
    E = EllipticCurve(0, cv.b, modulus=cv.fq.p)
    Gs = E(cv.gx, cv.gy)
    G = C.generator(cv)
    xy = lambda pt: (int(pt.x), int(pt.y))
    r = cv.fr.p
    scalars = [1, 2, 3, 0xFFFF, r - 1] + field_elems(r, 77, 6)
    pts = []
    for k in scalars:
        want = xy(k * Gs)
        assert C.scalar_mul(cv, k, G) == want
        pts.append(want)
    assert C.add(cv, pts[1], pts[2]) == xy(5 * Gs) and C.double(cv, pts[3]) == xy((2 * 0xFFFF) * Gs)
    assert C.add(cv, pts[4], G) is None                                    # (r - 1) G + G = O: the generator has order r
    coeffs = field_elems(r, 78, len(pts))
    acc = None
    for s, (k, _) in zip(coeffs, zip(scalars, pts)):
        acc = (s * k) % r if acc is None else (acc + s * k) % r
    want = xy(acc * Gs)
    assert C.msm_pippenger(cv, pts, coeffs) == want
    out, inf = K.msm_mont(cv, K.points_to_mont(cv, pts), K.fr_to_mont(cv, coeffs))
    assert not inf and K.points_from_mont(cv, out)[0] == want
    tau = 0xC0FFEE
    srs = K.points_from_mont(cv, K.srs_mont(cv, tau, 6))
    assert srs == [xy(pow(tau, i, r) * Gs) for i in range(6)]
