"""GPU parity: the HIP NTT (through the C-ABI) against the CPU oracle and the golden fixtures."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, coracle as K
from helpers import field_elems, digest, rand_fr

CURVES = [F.BN254, F.BLS12_381]
VARIANTS = [("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)]


@pytest.fixture(scope="module")
def ctxs():
    import zkt_plonk_amd as z
    c = {cv.name: z.Context(cv.name, 0) for cv in CURVES}
    yield c
    for x in c.values():
        x.close()


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_compiled_constants(cv, ctxs):
    for which, f in ((0, cv.fr), (1, cv.fq)):
        pr = ctxs[cv.name].debug_params(which)
        assert pr == dict(p=f.p, inv32=f.inv32, r=f.R, r2=f.R2)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_field_mul_bit_exact(cv, ctxs):
    p = cv.fr.p
    a = field_elems(p, 1, 4096) + [0, 1, p - 1, p - 1]
    b = field_elems(p, 2, 4096) + [5, p - 1, p - 1, 0]
    got = K.fr_from_mont(cv, ctxs[cv.name].debug_fr_mul(K.fr_to_mont(cv, a), K.fr_to_mont(cv, b)))
    assert got == [x * y % p for x, y in zip(a, b)]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_group_gen(cv, ctxs):
    for log_n in (0, 1, 10, 20, cv.fr.two_adicity):
        w = K.fr_from_mont(cv, ctxs[cv.name].group_gen(log_n).reshape(1, 4))[0]
        assert w == cv.fr.root_of_unity(1 << log_n)
    import zkt_plonk_amd as z
    with pytest.raises(z.ZktError) as e:
        ctxs[cv.name].group_gen(cv.fr.two_adicity + 1)
    assert e.value.code == 2


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ntt_golden_vectors(cv, ctxs, golden):
    p = cv.fr.p
    for e in golden[cv.name]["ntt"]:
        log_n = e["n"].bit_length() - 1
        x = K.fr_to_mont(cv, field_elems(p, e["seed"], e["in_len"]))
        for name, inv, cos in VARIANTS:
            y = K.fr_from_mont(cv, ctxs[cv.name].ntt(log_n, x, inverse=bool(inv), coset=bool(cos)))
            assert digest(y) == e[name + "_sha256"], (cv.name, e["n"], name)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_ntt_matches_oracle_every_size(cv, log_n, ctxs):
    rng = np.random.default_rng(1000 + log_n)
    n = 1 << log_n
    # ragged input (zero padded by the transform) and full input
    lens = sorted({n, max(1, n - 3), max(1, n // 4 + 3) if n >= 4 else n})
    if log_n >= 17:
        lens = [n // 4 + 3]   # the prover's shape: <= n/4 + 3 coefficients on the 4x domain
    if log_n == 20:
        lens = [n // 4 + 3, n]   # BASELINE.json configs[1]: the literal full-length 2^20 input, every coefficient compared
    for in_len in lens:
        x = rand_fr(rng, in_len)
        for name, inv, cos in VARIANTS:
            got = ctxs[cv.name].ntt(log_n, x, inverse=bool(inv), coset=bool(cos))
            want = K.ntt_mont(cv, log_n, inv, cos, x)
            assert np.array_equal(got, want), (cv.name, log_n, in_len, name)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ntt_empty_and_oversized_inputs(cv, ctxs):
    import zkt_plonk_amd as z
    ctx = ctxs[cv.name]
    out = ctx.ntt(4, np.zeros((0, 4), dtype=np.uint64))
    assert out.shape == (16, 4) and not out.any()
    with pytest.raises(z.ZktError) as e:
        ctx.ntt(3, np.zeros((9, 4), dtype=np.uint64))
    assert e.value.code == 2
    with pytest.raises(z.ZktError):
        ctx.ntt(cv.fr.two_adicity + 1 if cv.fr.two_adicity < 27 else 28, np.zeros((1, 4), dtype=np.uint64))


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_ntt_full_size_properties(cv, ctxs):
    """Size-independent properties at BASELINE sizes (2^20 and the 4n = 2^22 domain):
    round trips, linearity, and a spot check of the definition at a few output indices."""
    ctx = ctxs[cv.name]
    p = cv.fr.p
    rng = np.random.default_rng(99)
    for log_n in (20, 22):
        n = 1 << log_n
        x = rand_fr(rng, n)
        y = ctx.ntt(log_n, x)
        assert np.array_equal(ctx.ntt(log_n, y, inverse=True), x)
        yc = ctx.ntt(log_n, x, coset=True)
        assert np.array_equal(ctx.ntt(log_n, yc, inverse=True, coset=True), x)
        # definition spot check: X[k] = sum_j x_j w^(jk) for a sparse x
        xs = np.zeros((n, 4), dtype=np.uint64)
        idx = [0, 1, 12345 % n, n - 1]
        vals = field_elems(p, 31337, len(idx))
        xs[idx] = K.fr_to_mont(cv, vals)
        ys = ctx.ntt(log_n, xs)
        w = cv.fr.root_of_unity(n)
        for k in (0, 1, 777, n // 2 + 5, n - 1):
            want = sum(v * pow(w, j * k, p) for j, v in zip(idx, vals)) % p
            assert K.fr_from_mont(cv, ys[k:k + 1])[0] == want




@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_domain_and_commitment_mirrors_of_the_reference_seams(cv, ctxs):
    """GpuDomain ~ D: EvaluationDomain + EvaluationDomainExt (util.rs:27-140) and GpuKZG10 ~ PC: HomomorphicCommitment
    (commitment.rs:10-46), the two generic seams of ZKTPlonk (plonk.rs:39-52), against the oracle's Domain / commit."""
    import zkt_plonk_amd as z
    from oracle.ntt import Domain
    from oracle import curve as C
    ctx = ctxs[cv.name]
    p = cv.fr.p
    for num_coeffs in (1, 5, 8, 100, 1000):                      # D::new rounds up to a power of two
        d, o = z.GpuDomain(ctx, num_coeffs), Domain(cv.fr, num_coeffs)
        assert d.size() == o.size and d.log_size_of_group() == o.log_size
        assert K.fr_from_mont(cv, d.group_gen().reshape(1, 4)) == [o.group_gen]
        x = field_elems(p, 40 + num_coeffs, min(num_coeffs, o.size))
        xm = K.fr_to_mont(cv, x)
        assert K.fr_from_mont(cv, d.fft(xm)) == o.fft(x)
        assert K.fr_from_mont(cv, d.ifft(xm)) == o.ifft(x)
        assert K.fr_from_mont(cv, d.coset_fft(xm)) == o.coset_fft(x)
        assert K.fr_from_mont(cv, d.coset_ifft(xm)) == o.coset_ifft(x)
    with pytest.raises(z.ZktError) as e:
        z.GpuDomain(ctx, (1 << cv.fr.two_adicity) + 1)           # Error::InvalidEvalDomainSize (prove.rs:77-81)
    assert e.value.code == 2
    srs = K.srs_mont(cv, 0xC0DE, 300)
    pc = z.GpuKZG10(ctx, srs)
    coeffs = field_elems(p, 9, 257)
    xy, inf = pc.commit(K.fr_to_mont(cv, coeffs))
    assert (None if inf else K.points_from_mont(cv, xy.reshape(1, -1))[0]) == C.msm_naive(cv, K.points_from_mont(cv, srs[:257]), coeffs)
    with pytest.raises(z.ZktError) as e:
        pc.commit(np.zeros((301, 4), dtype=np.uint64))
    assert e.value.code == 5
    sc = field_elems(p, 10, 13)
    xy, inf = pc.multi_scalar_mul(srs[20:33], K.fr_to_mont(cv, sc))
    assert (None if inf else K.points_from_mont(cv, xy.reshape(1, -1))[0]) == C.msm_naive(cv, K.points_from_mont(cv, srs[20:33]), sc)
