"""Writers for the reference CLI's key files (oracle; test infrastructure only -- never imported by the product).

`serialize_to_file` (bin/src/parser.rs:14-22) = CanonicalSerialize::serialize_unchecked.  The reference tree holds no
key file and its serialisation code lives in third-party crates that are absent from /root/reference (ark-serialize
0.3, ark-poly-commit 0.3, ark-ec 0.3), so these writers restate the published derive rules and pin the product's
readers (zkt-plonk_amd/csrc/keyfile.hip) only through the round trip: "parity unpinned".

  usize / u64 -> 8 bytes LE;  Vec<T> -> u64 length + elements;  String -> Vec<u8>;  Option<T> -> tag byte + T
  Fp -> canonical value, little endian;  GroupAffine (unchecked = uncompressed) -> x, y with SWFlags in y's top bits
  (bit 6 = infinity; GroupAffine::zero() is (0, 1, infinity))
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

from .curve import Point
from .fields import Curve
from . import plonk as P


def _u64(v: int) -> bytes:
    return int(v).to_bytes(8, "little")


def _opt_usize(v: Optional[int]) -> bytes:
    return b"\x00" if v is None else b"\x01" + _u64(v)


def _fr(cv: Curve, v: int) -> bytes:
    return int(v % cv.fr.p).to_bytes(cv.fr.limbs64 * 8, "little")


def _g1_unchecked(cv: Curve, pt: Point) -> bytes:
    nb = cv.fq.limbs64 * 8
    if pt is None:   # GroupAffine::zero() = (0, 1) with the infinity flag
        y = bytearray(int(1).to_bytes(nb, "little"))
        y[-1] |= 0x40
        return bytes(nb) + bytes(y)
    return int(pt[0]).to_bytes(nb, "little") + int(pt[1]).to_bytes(nb, "little")


def _vec_g1(cv: Curve, pts: Sequence[Point]) -> bytes:
    return _u64(len(pts)) + b"".join(_g1_unchecked(cv, p) for p in pts)


def committer_key_bytes(cv: Curve, powers_of_g: Sequence[Point], powers_of_gamma_g: Sequence[Point] = (),
                        max_degree: Optional[int] = None) -> bytes:
    """ark-poly-commit 0.3 sonic_pc::CommitterKey as PC::trim(pp, 4n, 0, None) leaves it (plonk.rs:79-85): powers_of_g,
    powers_of_gamma_g, shifted_powers_of_g = None, shifted_powers_of_gamma_g = None, enforced_degree_bounds = None,
    max_degree."""
    out = _vec_g1(cv, powers_of_g) + _vec_g1(cv, powers_of_gamma_g)
    out += b"\x00" + b"\x00" + b"\x00"
    out += _u64(len(powers_of_g) - 1 if max_degree is None else max_degree)
    return out


def _labeled_poly(cv: Curve, label: str, coeffs: Sequence[int]) -> bytes:
    lb = label.encode()
    return (_u64(len(lb)) + lb + _u64(len(coeffs)) + b"".join(_fr(cv, c) for c in coeffs)
            + _opt_usize(None) + _opt_usize(None))        # degree_bound, hiding_bound (setup.rs:92-101: both None)


def prover_key_bytes(cv: Curve, pk: P.ProverKey) -> bytes:
    """plonk-core ProverKey<F> (keys/mod.rs:29-41): arith {q_m q_l q_r q_o q_c}, perm {sigma1..3}, lookup {q_lookup
    q_table}, each a LabeledPolynomial labelled as in setup.rs:92-101."""
    return b"".join(_labeled_poly(cv, k, pk.polys[k]) for k in P.PK_POLYS)


def verifier_key_bytes(cv: Curve, vk: P.VerifierKey) -> bytes:
    """plonk-core VerifierKey (keys/mod.rs:180-210): n, pi_roots, then the ten commitments."""
    out = _u64(vk.n) + _u64(len(vk.pi_roots)) + b"".join(_fr(cv, r) for r in vk.pi_roots)
    return out + b"".join(_g1_unchecked(cv, vk.commits[k]) for k in P.PK_POLYS)


EPK_ORDER = ("q_m_coset", "q_l_coset", "q_r_coset", "q_o_coset", "q_c_coset", "q_lookup", "q_lookup_coset", "q_table_coset",
             "sigma1", "sigma1_coset", "sigma2", "sigma2_coset", "sigma3", "sigma3_coset", "x_coset", "zh_coset", "l_1_coset")


def extended_prover_key_vectors(epk: P.ExtendedProverKey) -> Dict[str, Sequence[int]]:
    """The seventeen Vec<F> of plonk-core's ExtendedProverKey<F> by field name (keys/mod.rs:148-174; keys/arithmetic.rs:51-62,
    keys/lookup.rs:70-77, keys/permutation.rs:74-92)."""
    out = {k + "_coset": epk.cosets[k] for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup", "q_table", "sigma1", "sigma2",
                                                 "sigma3", "x", "zh", "l_1")}
    out.update(q_lookup=epk.q_lookup, sigma1=epk.sigma1, sigma2=epk.sigma2, sigma3=epk.sigma3)
    return out


def extended_prover_key_bytes(cv: Curve, epk: P.ExtendedProverKey) -> bytes:
    """ExtendedProverKey<F> as `serialize_to_file` writes it (bin/src/main.rs:108-109): the derive walks the fields in
    declaration order -- arith, lookup, perm, zh_coset, l_1_coset -- and every one is a Vec<F>."""
    vecs = extended_prover_key_vectors(epk)
    return b"".join(_u64(len(vecs[k])) + b"".join(_fr(cv, v) for v in vecs[k]) for k in EPK_ORDER)
