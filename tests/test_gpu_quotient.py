"""The fused quotient kernel on its own (quotient_poly.rs:98-224): point-by-point equality with the oracle
on witness cosets that satisfy nothing, so no term can hide behind the vanishing of another; both public-input paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, plonk as P, coracle as K
from helpers import field_elems

CURVES = [F.BN254, F.BLS12_381]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("gates,table_size", [(100, 16), (4000, 256)])
def test_quotient_kernel_equals_the_oracle_pointwise(cv, gates, table_size):
    import zkt_plonk_amd as z
    p = cv.fr.p
    cs = P.synthetic_circuit(cv, gates, table_size, seed=gates + 1)
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    srs_arr = K.srs_mont(cv, 0x51DE + gates, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx = z.Context(cv.name, 0)
    try:
        ctx.srs_load(srs_arr)
        z.GpuProver(ctx, log_n, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64)
                                 for k in z.PK_ORDER})
        ch = K.fr_to_mont(cv, field_elems(p, 900 + gates, 5))
        wit = {k: K.fr_to_mont(cv, field_elems(p, 1000 + i, 4 * n)) for i, k in enumerate(K.WIT_ORDER)}
        cos = {k: K.fr_to_mont(cv, v) for k, v in epk.cosets.items()}
        want = K.quotient_evals(cv, log_n, ch, cos, wit)
        got = ctx.debug_quotient(n, ch, [wit[k] for k in K.WIT_ORDER])
        assert np.array_equal(got, want)
        # few public inputs: the kernel evaluates PI(X) from rotations of the l1 coset instead of reading a vector
        for pos in ([0], [0, 3, n - 1], list(range(5, 21))):
            vals = field_elems(p, 77 + len(pos), len(pos))
            pi_ev = [0] * n
            for k, v in zip(pos, vals):
                pi_ev[k] = v
            pi_poly = K.ntt_mont(cv, log_n, True, False, K.fr_to_mont(cv, pi_ev))
            w2 = dict(wit, pi=K.ntt_mont(cv, log_n + 2, False, True, pi_poly))
            want2 = K.quotient_evals(cv, log_n, ch, cos, w2)
            got2 = ctx.debug_quotient(n, ch, [None if k == "pi" else wit[k] for k in K.WIT_ORDER], pos, K.fr_to_mont(cv, vals))
            assert np.array_equal(got2, want2), len(pos)
            assert np.array_equal(ctx.debug_quotient(n, ch, [w2[k] for k in K.WIT_ORDER]), want2)
        with pytest.raises(z.ZktError):
            ctx.debug_quotient(n, ch, [None if k == "pi" else wit[k] for k in K.WIT_ORDER], [n], K.fr_to_mont(cv, [1]))
    finally:
        ctx.close()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("gates,table_size", [(100, 16), (5000, 256), (70000, 1024)])
def test_grand_products_and_opening_witness_alone(cv, gates, table_size):
    """Rows a8 / a9 / a12 / a13 on their own: the permutation and lookup grand products (permutation/mod.rs:181-254,
    lookup/mod.rs:94-151) on arbitrary vectors -- nothing has to satisfy anything, so a wrong term cannot hide behind a
    ratio of one -- the evaluation and linear-combination kernels of the linearisation (linearization_poly.rs:55-121) and
    the KZG witness polynomial (kzg10::compute_witness_polynomial), each against the CPU oracle
    element by element; then the lookup product on vectors of the prover's own shape (sorted halves of a real
    combine_split), whose long runs of ratio one are what the Lagrange-basis commitment of z2 relies on."""
    import zkt_plonk_amd as z
    p = cv.fr.p
    cs = P.synthetic_circuit(cv, gates, table_size, seed=gates + 7)
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    srs_arr = K.srs_mont(cv, 0x77AA + gates, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    ctx = z.Context(cv.name, 0)
    try:
        ctx.srs_load(srs_arr)
        z.GpuProver(ctx, log_n, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), dtype=np.uint64)
                                 for k in z.PK_ORDER})
        ch = K.fr_to_mont(cv, field_elems(p, 300 + gates, 4))
        vec = [K.fr_to_mont(cv, field_elems(p, 400 + i, n)) for i in range(7)]
        sig = [K.ntt_mont(cv, log_n, False, False, np.concatenate([K.fr_to_mont(cv, pk.polys[k]),
                                                                    np.zeros((n - len(pk.polys[k]), 4), dtype=np.uint64)]))
               for k in ("sigma1", "sigma2", "sigma3")]
        z1, z2 = ctx.debug_grand_products(n, ch, vec)
        assert np.array_equal(z1, K.z1_evals(cv, log_n, ch[0], ch[1], vec[0], vec[1], vec[2], *sig))
        assert np.array_equal(z2, K.z2_evals(cv, log_n, ch[2], ch[3], vec[3], vec[4], vec[5], vec[6]))
        # the prover's own shape: f = q_lookup . c, t padded with zeros, h1 / h2 = combine_split
        a, b, c = cs.wire_evals(n)
        ql = K.fr_from_mont(cv, K.ntt_mont(cv, log_n, False, False, np.concatenate(
            [K.fr_to_mont(cv, pk.polys["q_lookup"]), np.zeros((n - len(pk.polys["q_lookup"]), 4), dtype=np.uint64)])))
        f = [x * y % p for x, y in zip(ql, c)]
        t = list(cs.table) + [0] * (n - len(cs.table))
        h1, h2 = K.combine_split(K.fr_to_mont(cv, t), K.fr_to_mont(cv, f))
        vec2 = [K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, f), K.fr_to_mont(cv, t), h1, h2]
        z1, z2 = ctx.debug_grand_products(n, ch, vec2)
        assert np.array_equal(z1, K.z1_evals(cv, log_n, ch[0], ch[1], vec2[0], vec2[1], vec2[2], *sig))
        want2 = K.z2_evals(cv, log_n, ch[2], ch[3], vec2[3], vec2[4], vec2[5], vec2[6])
        assert np.array_equal(z2, want2)
        runs = 1 + int(np.count_nonzero(np.any(want2[1:] != want2[:-1], axis=1)))
        assert runs <= 4 * (len(cs.table) + n // 16 + 2)                  # piecewise constant: what makes its commitment cheap
        # row a13 alone: evaluations at a point each and a linear combination of polynomials of the prover's lengths
        # (linearization_poly.rs:55-121), against Horner's rule and the plain sum on big integers
        lens = sorted({l for l in (n + 8, n + 3, n, n // 2 + 1, 2049, 2048, 5, 1) if l <= n + 8}, reverse=True)[:7]
        polys = [field_elems(p, 700 + i, ln) for i, ln in enumerate(lens)]
        pts = field_elems(p, 710, len(lens))
        pts[0] = 0                                                      # p(0) = the constant term
        scs = field_elems(p, 711, len(lens))
        out_len = n + 8
        ev, lc = ctx.debug_eval_lincomb([K.fr_to_mont(cv, q) for q in polys], K.fr_to_mont(cv, pts), K.fr_to_mont(cv, scs), out_len)
        want_ev = []
        for q, x in zip(polys, pts):
            acc = 0
            for coef in reversed(q):
                acc = (acc * x + coef) % p
            want_ev.append(acc)
        assert K.fr_from_mont(cv, ev) == want_ev
        want_lc = [0] * out_len
        for q, sc in zip(polys, scs):
            for i, coef in enumerate(q):
                want_lc[i] = (want_lc[i] + sc * coef) % p
        assert K.fr_from_mont(cv, lc) == want_lc
        # the witness polynomial of an opening: full length, lengths either side of a workgroup's 512 elements, short ones
        for ln in sorted({l for l in (n + 3, n + 8, n // 2 + 1, 1025, 513, 512, 5, 2) if l <= n + 8}):
            poly = K.fr_to_mont(cv, field_elems(p, 900 + ln, ln))
            zz = K.fr_to_mont(cv, field_elems(p, 901, 1))[0]
            assert np.array_equal(ctx.debug_open_witness(poly, zz), K.div_linear(cv, poly, zz))
    finally:
        ctx.close()
