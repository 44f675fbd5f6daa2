"""Commitments of evaluation vectors through the Lagrange-basis key (csrc/lagrange.hip, include/zkt_plonk.h): the point must
be the one PC::commit gives for poly_from_evals(evals) + add_blinders_to_poly (prove.rs:166-180, 472-483), which the CPU
oracle computes the reference's way: inverse transform, trim, blinders, one MSM over the coefficients."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, plonk as P, coracle as K
from helpers import field_elems

CURVES = [F.BN254, F.BLS12_381]


def _oracle_commit(cv, log_n, srs_arr, evals, blinders):
    """util.rs:63-86 + prove.rs:472-483 + commitment.rs:24-46 on the CPU."""
    n = 1 << log_n
    p = cv.fr.p
    coeffs = K.fr_from_mont(cv, K.ntt_mont(cv, log_n, True, False, K.fr_to_mont(cv, evals)))
    while coeffs and coeffs[-1] == 0:
        coeffs.pop()                                  # DensePolynomial::from_coefficients_vec
    coeffs = coeffs + list(blinders)                  # coeffs.extend(blinders)
    for i, b in enumerate(blinders):                  # coeffs[i] -= blinder[i]
        coeffs[i] = (coeffs[i] - b) % p
    if not coeffs:
        return None
    out, inf = K.msm_mont(cv, srs_arr[:len(coeffs)], K.fr_to_mont(cv, coeffs))
    return None if inf else K.points_from_mont(cv, out)[0]


def _vectors(cv, log_n):
    n = 1 << log_n
    p = cv.fr.p
    dense = field_elems(p, 31 + log_n, n)
    vals = field_elems(p, 77, 9)
    runs = []                                         # piecewise constant, a handful of runs (what h1 / h2 / z2 look like)
    cuts = sorted(set([0, 1, 2, n // 3, n // 3 + 1, n // 2, n - 2]) & set(range(n)))
    for i in range(n):
        runs.append(vals[sum(1 for c in cuts if c <= i) % 9])
    table_like = field_elems(p, 5, min(7, n // 2)) + [0] * (n - min(7, n // 2))      # t: values, then zeros
    # polynomials whose top coefficients vanish: the blinders land below X^n (degenerate lengths n - 1, n - 2, 1, 0)
    def evals_of(coeffs):
        return K.fr_from_mont(cv, K.ntt_mont(cv, log_n, False, False, K.fr_to_mont(cv, coeffs + [0] * (n - len(coeffs)))))
    short1 = evals_of(field_elems(p, 91, n - 1))
    short2 = evals_of(field_elems(p, 92, n - 2))
    return {"dense": dense, "runs": runs, "table": table_like, "zeros": [0] * n, "const": [5] * n, "deg_n-2": short1,
            "deg_n-3": short2, "x": evals_of([0, 1])}


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [3, 6, 11])
def test_commit_evals_equals_the_coefficient_commitment(cv, log_n):
    import zkt_plonk_amd as z
    n = 1 << log_n
    p = cv.fr.p
    cs = P.synthetic_circuit(cv, n - 3, 4, seed=log_n, n_public=2)
    assert cs.circuit_bound() == n
    srs_arr = K.srs_mont(cv, 0x1A6 + log_n, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx = z.Context(cv.name, 0)
    try:
        ctx.srs_load(srs_arr)
        z.GpuProver(ctx, log_n, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64)
                                 for k in z.PK_ORDER})
        d_ev = ctx.alloc(n * 32)
        bl_all = field_elems(p, 4242, 3)
        for name, ev in _vectors(cv, log_n).items():
            ctx.upload(d_ev, K.fr_to_mont(cv, ev))
            for k in range(4):
                bl = bl_all[:k]
                want = _oracle_commit(cv, log_n, srs_arr, ev, bl)
                for path in (0, 1):
                    out, inf = ctx.commit_evals_dev(d_ev, K.fr_to_mont(cv, bl) if k else None, path)
                    got = None if inf else K.points_from_mont(cv, out)[0]
                    assert got == want, (name, k, path)
        info = ctx.lagrange_info()
        assert info["log_n"] == log_n and info["bases"] == n + 8
        ctx.free(d_ev)
    finally:
        ctx.close()


def test_short_key_and_switch_fall_back_to_coefficients():
    """A key with exactly n + 3 powers carries three blinder bases; one with n powers none at all (the table is not built and
    path 1 says so); zkt_ctx_set_lagrange(0) keeps the prover on the coefficient route with the same bytes."""
    import zkt_plonk_amd as z
    cv = F.BN254
    log_n = 6
    n = 1 << log_n
    p = cv.fr.p
    cs = P.synthetic_circuit(cv, n - 3, 4, seed=3)
    srs_arr = K.srs_mont(cv, 0x77, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    pkm = {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64) for k in z.PK_ORDER}
    ev = field_elems(p, 9, n)
    bl = field_elems(p, 10, 3)
    ctx = z.Context(cv.name, 0)
    try:
        ctx.srs_load(srs_arr[:n + 3])
        z.GpuProver(ctx, log_n, pkm)
        d_ev = ctx.alloc(n * 32)
        ctx.upload(d_ev, K.fr_to_mont(cv, ev))
        want = _oracle_commit(cv, log_n, srs_arr, ev, bl)
        out, inf = ctx.commit_evals_dev(d_ev, K.fr_to_mont(cv, bl), 1)
        assert K.points_from_mont(cv, out)[0] == want
        assert ctx.lagrange_info()["bases"] == n + 3
        ctx.srs_load(srs_arr[:n])
        with pytest.raises(z.ZktError):
            ctx.commit_evals_dev(d_ev, None, 1)
        out, inf = ctx.commit_evals_dev(d_ev, None, 0)
        assert K.points_from_mont(cv, out)[0] == _oracle_commit(cv, log_n, srs_arr, ev, [])
        ctx.free(d_ev)
    finally:
        ctx.close()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_proof_bytes_do_not_depend_on_the_commitment_route(cv):
    import zkt_plonk_amd as z
    from test_gpu_prove import _gpu_prove
    cs = P.synthetic_circuit(cv, 900, 64, seed=12)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0xBEE, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 555, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    ctx = z.Context(cv.name, 0)
    try:
        got_on = _gpu_prove(z, ctx, cv, cs, pk, vk, srs_arr, blinders)
        assert ctx.lagrange_info()["log_n"] == n.bit_length() - 1
        ctx.set_lagrange(False)
        assert ctx.lagrange_info()["log_n"] == -1
        got_off = _gpu_prove(z, ctx, cv, cs, pk, vk, srs_arr, blinders)
        assert got_on == want and got_off == want
    finally:
        ctx.close()
