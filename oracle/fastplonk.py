"""Array form of the oracle prover (oracle; test infrastructure only -- never imported by the product).

Same algorithm, same order and same transcript as ``oracle/plonk.py`` (which restates
plonk-core/src/proof_system/prove.rs:59-470 and setup.rs:42-166 function by function on Python
integers), but every vector is a numpy array of arkworks Montgomery limbs and every O(n) loop
runs in the C++ restatement (oracle/coracle.cpp), so that whole proofs at n = 2^14 .. 2^22 are
available as the byte-exact check of the GPU path and as bench.py's timed CPU baseline.
tests/test_coracle.py pins it to ``oracle/plonk.py`` byte for byte at n <= 4096.

Scalars (challenges, blinders, evaluations) are canonical Python ints at this level; they cross
to C as 4 Montgomery limbs.  The Fiat-Shamir transcript is the Python one (oracle/transcript.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import coracle as K
from . import curve as C
from . import plonk as P
from .fields import Curve, K1, K2

Z4 = np.zeros((0, 4), dtype=np.uint64)


def _m(cv: Curve, v: int) -> np.ndarray:
    """canonical int -> (4,) Montgomery limbs"""
    return K.fr_to_mont(cv, [v % cv.fr.p])[0]


def _i(cv: Curve, limbs) -> int:
    return K.fr_from_mont(cv, np.asarray(limbs, dtype=np.uint64).reshape(1, 4))[0]


def _trim(arr: np.ndarray) -> np.ndarray:
    return arr[:K.trim_len(arr)]


def _pad(arr: np.ndarray, n: int) -> np.ndarray:
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    assert arr.shape[0] <= n
    if arr.shape[0] == n:
        return arr
    out = np.zeros((n, 4), dtype=np.uint64)
    out[:arr.shape[0]] = arr
    return out


@dataclass
class FastKeys:
    """ProverKey + ExtendedProverKey + VerifierKey (keys/mod.rs:29-201) on arrays."""
    n: int
    log_n: int
    pk: Dict[str, np.ndarray]            # coefficient form, trailing zeros stripped
    epk: Dict[str, np.ndarray]           # 13 vectors of 4n coset evaluations
    sigma: List[np.ndarray]              # n evaluations each
    q_lookup_ev: np.ndarray
    commits: Dict[str, C.Point] = field(default_factory=dict)

    def verifier_key(self, cv: Curve, pi_pos: Sequence[int]) -> P.VerifierKey:
        w = cv.fr.root_of_unity(self.n)
        return P.VerifierKey(self.n, [pow(w, i, cv.fr.p) for i in sorted(pi_pos)], dict(self.commits))


def commit(cv: Curve, srs_arr: np.ndarray, poly: np.ndarray) -> C.Point:
    """commitment.rs:24 / kzg10::commit = MSM(powers_of_g[..len], coeffs)."""
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    if poly.shape[0] > srs_arr.shape[0]:
        raise ValueError("TooManyCoefficients")
    if poly.shape[0] == 0:
        return None
    out, inf = K.msm_mont(cv, srs_arr[:poly.shape[0]], poly, True)
    return None if inf else K.points_from_mont(cv, out)[0]


def extend_prover_key(cv: Curve, log_n: int, pk: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """keys/mod.rs:78-146."""
    n = 1 << log_n
    cos = {k: K.ntt_mont(cv, log_n + 2, False, True, pk[k]) for k in
           ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup", "q_table", "sigma1", "sigma2", "sigma3")}
    one = _m(cv, 1)
    cos["x"] = K.ntt_mont(cv, log_n + 2, False, True, np.stack([_m(cv, 0), one]))            # mod.rs:110-113
    zh = np.zeros((n + 1, 4), dtype=np.uint64)
    zh[0] = _m(cv, cv.fr.p - 1)
    zh[n] = one
    cos["zh"] = K.ntt_mont(cv, log_n + 2, False, True, zh)                                    # mod.rs:115-117
    l1_ev = np.zeros((n, 4), dtype=np.uint64)
    l1_ev[0] = one
    cos["l_1"] = K.ntt_mont(cv, log_n + 2, False, True, _trim(K.ntt_mont(cv, log_n, True, False, l1_ev)))  # :119-120
    return cos


def setup(cv: Curve, srs_arr: np.ndarray, log_n: int, evals: Dict[str, np.ndarray], commitments: bool = True) -> FastKeys:
    """proof_system/setup.rs:42-166 from the ten padded evaluation vectors (oracle.plonk.setup_evals)."""
    n = 1 << log_n
    ev = {k: _pad(evals[k], n) for k in P.PK_POLYS}
    pk = {k: _trim(K.ntt_mont(cv, log_n, True, False, ev[k])) for k in P.PK_POLYS}
    commits = {k: commit(cv, srs_arr, pk[k]) for k in P.PK_POLYS} if commitments else {}
    epk = extend_prover_key(cv, log_n, pk)
    return FastKeys(n, log_n, pk, epk, [ev["sigma1"], ev["sigma2"], ev["sigma3"]], ev["q_lookup"], commits)


def add_blinders(cv: Curve, poly: np.ndarray, blinders: Sequence[int]) -> np.ndarray:
    """prove.rs:472-483 on the trimmed coefficient vector."""
    k = len(blinders)
    b = K.fr_to_mont(cv, [x % cv.fr.p for x in blinders])
    out = np.zeros((poly.shape[0] + k, 4), dtype=np.uint64)
    out[:poly.shape[0]] = poly
    out[poly.shape[0]:] = b
    out[:k] = K.vec_op(cv, "sub", out[:k], b)
    return out


def _scaled_sum(cv: Curve, polys, scalars) -> np.ndarray:
    """sum_k s_k p_k as a trimmed DensePolynomial (poly_add / poly_scale of oracle/plonk.py)."""
    L = max((q.shape[0] for q in polys), default=0)
    if L == 0:
        return Z4
    return _trim(K.lincomb(cv, polys, K.fr_to_mont(cv, [s % cv.fr.p for s in scalars]), L))


def kzg_open(cv: Curve, srs_arr, polys, point: int, eta: int) -> C.Point:
    """SonicKZG10::open -> kzg10::open (oracle.plonk.kzg_open)."""
    p = cv.fr.p
    comb = _scaled_sum(cv, polys, [pow(eta, k, p) for k in range(len(polys))])
    if comb.shape[0] < 2:
        return None
    return commit(cv, srs_arr, _trim(K.div_linear(cv, comb, _m(cv, point))))


def prove(cv: Curve, srs_arr: np.ndarray, keys: FastKeys, a, b, c, table, pi: Dict[int, int], transcript,
          blinders: Sequence[int], trace: Optional[dict] = None) -> bytes:
    """proof_system/prove.rs:59-470 (the array twin of oracle.plonk.prove).  a, b, c: wire evaluations (<= n rows,
    Montgomery); table: the LookupTable's distinct values in insertion order; pi: position -> canonical value;
    transcript: already seeded.  Returns the CanonicalSerialize bytes of the Proof."""
    p = cv.fr.p
    n, log_n = keys.n, keys.log_n
    assert len(blinders) == P.NUM_BLINDERS
    bl, off = {}, 0
    for name, k in P.BLINDER_LAYOUT:
        bl[name] = [x % p for x in blinders[off:off + k]]
        off += k
    ifft = lambda ev: _trim(K.ntt_mont(cv, log_n, True, False, ev))

    transcript.append_scalars("pi", [pi[k] for k in sorted(pi)])                         # prove.rs:110

    # round 1 (prove.rs:116-140)
    a_ev, b_ev, c_ev = _pad(a, n), _pad(b, n), _pad(c, n)
    a_poly = add_blinders(cv, ifft(a_ev), bl["a"])
    b_poly = add_blinders(cv, ifft(b_ev), bl["b"])
    c_poly = add_blinders(cv, ifft(c_ev), bl["c"])
    com = {"a": commit(cv, srs_arr, a_poly), "b": commit(cv, srs_arr, b_poly), "c": commit(cv, srs_arr, c_poly)}
    for k in ("a", "b", "c"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 2 (prove.rs:145-185)
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    assert table.shape[0] < n
    t_ev = _pad(table, n)
    t_poly = ifft(t_ev)
    f_ev = K.vec_op(cv, "mul", keys.q_lookup_ev, c_ev)                                  # prove.rs:157-161
    h1_ev, h2_ev = K.combine_split(t_ev, f_ev)                                          # prove.rs:163
    assert h1_ev.shape[0] == n and h2_ev.shape[0] == n
    h1_poly = add_blinders(cv, ifft(h1_ev), bl["h1"])
    h2_poly = add_blinders(cv, ifft(h2_ev), bl["h2"])
    com.update(t=commit(cv, srs_arr, t_poly), h1=commit(cv, srs_arr, h1_poly), h2=commit(cv, srs_arr, h2_poly))
    for k in ("t", "h1", "h2"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 3 (prove.rs:190-255)
    beta = transcript.challenge_scalar("beta")
    gamma = transcript.challenge_scalar("gamma")
    delta = transcript.challenge_scalar("delta")
    epsilon = transcript.challenge_scalar("epsilon")
    assert len({beta, gamma, delta, epsilon}) == 4, "challenges must be different"
    z1_ev = K.z1_evals(cv, log_n, _m(cv, beta), _m(cv, gamma), a_ev, b_ev, c_ev, *keys.sigma)
    z1_poly = add_blinders(cv, ifft(z1_ev), bl["z1"])
    z2_ev = K.z2_evals(cv, log_n, _m(cv, delta), _m(cv, epsilon), f_ev, t_ev, h1_ev, h2_ev)
    z2_poly = add_blinders(cv, ifft(z2_ev), bl["z2"])
    com.update(z1=commit(cv, srs_arr, z1_poly), z2=commit(cv, srs_arr, z2_poly))
    transcript.append_commitment("z1_commit", com["z1"])
    transcript.append_commitment("z2_commit", com["z2"])

    # round 4 (prove.rs:258-313)
    pi_ev = np.zeros((n, 4), dtype=np.uint64)
    if pi:
        pos = sorted(pi)
        pi_ev[pos] = K.fr_to_mont(cv, [pi[k] for k in pos])
    pi_poly = ifft(pi_ev)
    alpha = transcript.challenge_scalar("alpha")
    ch = (alpha, beta, gamma, delta, epsilon)
    src = dict(z1=z1_poly, z2=z2_poly, a=a_poly, b=b_poly, c=c_poly, pi=pi_poly, t=t_poly, h1=h1_poly, h2=h2_poly)
    cosets = {k: K.ntt_mont(cv, log_n + 2, False, True, v) for k, v in src.items()}
    q_ev = K.quotient_evals(cv, log_n, K.fr_to_mont(cv, list(ch)), keys.epk, cosets)
    del cosets
    q_poly = _trim(K.ntt_mont(cv, log_n + 2, True, True, q_ev))
    if q_poly.shape[0] < 2 * (n + 2):
        raise IndexError("quotient polynomial too short to split (prove.rs:287-292 would panic)")
    q_lo = _trim(q_poly[:n + 2])
    q_mid = _trim(q_poly[n + 2:2 * (n + 2)]).copy()
    q_hi = _trim(q_poly[2 * (n + 2):]).copy()
    b0, b1 = _m(cv, bl["q"][0]), _m(cv, bl["q"][1])
    q_lo = np.concatenate([q_lo, b0[None]])                                              # prove.rs:297
    if not q_mid.shape[0] or not q_hi.shape[0]:
        raise IndexError("empty quotient chunk (prove.rs:298/300 would panic)")
    q_mid[0] = K.vec_op(cv, "sub", q_mid[:1], b0[None])[0]                               # prove.rs:298
    q_mid = np.concatenate([q_mid, b1[None]])                                            # prove.rs:299
    q_hi[0] = K.vec_op(cv, "sub", q_hi[:1], b1[None])[0]                                 # prove.rs:300
    com.update(q_lo=commit(cv, srs_arr, q_lo), q_mid=commit(cv, srs_arr, q_mid), q_hi=commit(cv, srs_arr, q_hi))
    for k in ("q_lo", "q_mid", "q_hi"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 5 (prove.rs:318-451, linearization_poly.rs:19-121)
    xi = transcript.challenge_scalar("xi")
    w = cv.fr.root_of_unity(n)
    shifted = xi * w % p
    PK = keys.pk
    ev_at = lambda poly, pt: _i(cv, K.poly_eval(cv, poly, _m(cv, pt)))
    ev = P.ProofEvaluations(
        a=ev_at(a_poly, xi), b=ev_at(b_poly, xi), c=ev_at(c_poly, xi),
        sigma1=ev_at(PK["sigma1"], xi), sigma2=ev_at(PK["sigma2"], xi), z1_next=ev_at(z1_poly, shifted),
        q_lookup=ev_at(PK["q_lookup"], xi), t=ev_at(t_poly, xi), t_next=ev_at(t_poly, shifted),
        z2_next=ev_at(z2_poly, shifted), h1_next=ev_at(h1_poly, shifted), h2=ev_at(h2_poly, xi))
    zh_eval = (pow(xi, n, p) - 1) % p
    l1 = P.lagrange_evaluation(cv, n, 1, zh_eval, xi)
    arith = _scaled_sum(cv, [PK["q_m"], PK["q_l"], PK["q_r"], PK["q_o"], PK["q_c"]],
                        [ev.a * ev.b, ev.a, ev.b, ev.c, 1])                              # keys/arithmetic.rs:37-46
    bxi = beta * xi % p
    alpha2 = alpha * alpha % p
    alpha3 = alpha2 * alpha % p
    alpha4 = alpha3 * alpha % p
    s_z1 = (alpha * (bxi + ev.a + gamma) % p * (bxi * K1 + ev.b + gamma) % p * (bxi * K2 + ev.c + gamma) + l1 * alpha2) % p
    s_s3 = (-alpha) * beta % p * ev.z1_next % p * (beta * ev.sigma1 + ev.a + gamma) % p * (beta * ev.sigma2 + ev.b + gamma) % p
    perm = _scaled_sum(cv, [z1_poly, PK["sigma3"]], [s_z1, s_s3])                        # keys/permutation.rs:34-69
    opd = (delta + 1) % p
    eopd = epsilon * opd % p
    s_z2 = (alpha3 * opd % p * (epsilon + ev.q_lookup * ev.c) % p * (eopd + ev.t + delta * ev.t_next) + alpha4 * l1) % p
    s_h1 = (-alpha3) * ev.z2_next % p * (eopd + ev.h2 + delta * ev.h1_next) % p
    s_qt = alpha4 * alpha % p * ev.t % p
    lookup = _scaled_sum(cv, [z2_poly, h1_poly, PK["q_table"]], [s_z2, s_h1, s_qt])      # keys/lookup.rs:29-65
    xn2 = (zh_eval + 1) * xi % p * xi % p                                                # linearization_poly.rs:103
    qt = _scaled_sum(cv, [q_hi, q_mid], [xn2, 1])
    qt = _scaled_sum(cv, [qt, q_lo], [xn2, 1])
    qt = _scaled_sum(cv, [qt], [(-zh_eval) % p])
    r_poly = _scaled_sum(cv, [arith, perm, lookup, qt], [1, 1, 1, 1])
    for name, key in (("a_eval", "a"), ("b_eval", "b"), ("c_eval", "c"), ("sigma1_eval", "sigma1"),
                      ("sigma2_eval", "sigma2"), ("z1_next_eval", "z1_next"), ("q_lookup_eval", "q_lookup"),
                      ("t_eval", "t"), ("t_next_eval", "t_next"), ("z2_next_eval", "z2_next"),
                      ("h1_next_eval", "h1_next"), ("h2_eval", "h2")):
        transcript.append_scalar(name, getattr(ev, key))
    eta = transcript.challenge_scalar("eta")
    aw = kzg_open(cv, srs_arr, [r_poly, a_poly, b_poly, c_poly, PK["sigma1"], PK["sigma2"], PK["q_lookup"], t_poly,
                                h2_poly], xi, eta)
    saw = kzg_open(cv, srs_arr, [z1_poly, z2_poly, t_poly, h1_poly], shifted, eta)
    if trace is not None:
        trace.update(challenges=dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, epsilon=epsilon, xi=xi, eta=eta),
                     polys=dict(a=a_poly, b=b_poly, c=c_poly, t=t_poly, h1=h1_poly, h2=h2_poly, z1=z1_poly, z2=z2_poly,
                                q=q_poly, q_lo=q_lo, q_mid=q_mid, q_hi=q_hi, r=r_poly),
                     evals=dict(z1=z1_ev, z2=z2_ev, q=q_ev, f=f_ev, h1=h1_ev, h2=h2_ev))
    return P.Proof(commits={k: com[k] for k in P.Proof.COMMIT_ORDER}, aw_opening=aw, saw_opening=saw,
                   evaluations=ev).serialize(cv)
