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
