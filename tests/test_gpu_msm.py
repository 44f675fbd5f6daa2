"""GPU parity: the HIP KZG/G1 MSM (through the C-ABI) against the CPU oracle and golden fixtures."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, curve as C, coracle as K
from helpers import field_elems, unhex_point, rand_fr

CURVES = [F.BN254, F.BLS12_381]


@pytest.fixture(scope="module")
def ctxs():
    import zkt_plonk_amd as z
    c = {cv.name: z.Context(cv.name, 0) for cv in CURVES}
    yield c
    for x in c.values():
        x.close()


def _pt(cv, out, inf):
    return None if inf else K.points_from_mont(cv, out)[0]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_srs_generate_matches_oracle(cv, ctxs, golden):
    ctx = ctxs[cv.name]
    tau = int(golden[cv.name]["tau"], 16)
    ctx.srs_generate(tau, 600)
    got = ctx.srs_download(0, 600)
    want = K.srs_mont(cv, tau, 600)
    assert np.array_equal(got, want)
    assert K.points_from_mont(cv, got[:4]) == [unhex_point(p) for p in golden[cv.name]["srs_first"]]


def test_msm_reproduces_the_public_bn254_doubling_vector(ctxs):
    """EIP-196's bn256Add(G, G) -- a known answer that comes from neither the reference nor this repository -- through
    the device MSM: with tau = 1 every SRS power is G, so <(1, 1), SRS> = <(2), SRS> = 2 G."""
    from test_oracle_primitives import BN254_2G
    cv = F.BN254
    ctx = ctxs[cv.name]
    ctx.srs_generate(1, 64)
    assert K.points_from_mont(cv, ctx.srs_download(0, 2)) == [C.generator(cv)] * 2
    for sc in ([2], [1, 1], [0, 2, 0], [cv.fr.p - 1, 3], [1] * 2 + [0] * 40):
        out, inf = ctx.msm(K.fr_to_mont(cv, sc))
        assert _pt(cv, out, inf) == BN254_2G, sc
        out, inf = ctx.msm(K.ints_to_limbs(sc, 4), montgomery=False)
        assert _pt(cv, out, inf) == BN254_2G, sc


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_msm_golden_vectors(cv, ctxs, golden):
    ctx = ctxs[cv.name]
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    p = cv.fr.p
    ctx.srs_load(K.srs_mont(cv, tau, 1000))
    info = ctx.msm_info()
    assert info["srs_count"] == 1000 and info["window_bits"] >= 8
    for e in g["msm"]:
        n = e["n"]
        if e["seed"] is None:
            sc = [int(s, 16) for s in e["scalars"]]
        else:
            sc = field_elems(p, e["seed"], n)
            if n >= 31:
                sc[0], sc[1], sc[2], sc[5] = 0, 1, p - 1, 0
        out, inf = ctx.msm(K.fr_to_mont(cv, sc))
        assert _pt(cv, out, inf) == unhex_point(e["result"]), (cv.name, n)
        out, inf = ctx.msm(K.ints_to_limbs(sc, 4), montgomery=False)  # canonical-bigint entry (commitment.rs:36-42)
        assert _pt(cv, out, inf) == unhex_point(e["result"]), (cv.name, n, "canonical")


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_msm_edge_cases(cv, ctxs, golden):
    import zkt_plonk_amd as z
    ctx = ctxs[cv.name]
    tau = int(golden[cv.name]["tau"], 16)
    p = cv.fr.p
    srs = K.srs_mont(cv, tau, 300)
    srs[7] = 0          # a base at infinity is skipped
    srs[9] = srs[8]     # duplicated base: P + P must take the doubling path
    ctx.srs_load(srs)
    pts = K.points_from_mont(cv, srs)
    # empty input -> identity
    out, inf = ctx.msm(np.zeros((0, 4), dtype=np.uint64))
    assert inf and not out.any()
    # all-zero scalars -> identity
    out, inf = ctx.msm(np.zeros((300, 4), dtype=np.uint64))
    assert inf
    # equal scalars on the duplicated bases, scalar on the infinity base, P + (-P)
    sc = [0] * 300
    sc[7], sc[8], sc[9] = 5, 3, 3
    out, inf = ctx.msm(K.fr_to_mont(cv, sc))
    assert _pt(cv, out, inf) == C.scalar_mul(cv, 6, pts[8])
    sc = [0] * 300
    sc[8], sc[9] = 11, p - 11
    out, inf = ctx.msm(K.fr_to_mont(cv, sc))
    assert inf
    # many identical small scalars (one crowded bucket) and extreme scalars
    sc = [1] * 300
    sc[0], sc[1], sc[2] = p - 1, p - 2, (1 << 200) + 12345
    out, inf = ctx.msm(K.fr_to_mont(cv, sc))
    want, winf = K.msm_mont(cv, srs, K.fr_to_mont(cv, sc))
    assert _pt(cv, out, inf) == _pt(cv, want, winf)
    # base_offset (kzg10 skip_leading_zeros) and the TooManyCoefficients error
    sc = field_elems(p, 4, 50)
    out, inf = ctx.msm(K.fr_to_mont(cv, sc), base_offset=100)
    want, winf = K.msm_mont(cv, srs[100:150], K.fr_to_mont(cv, sc))
    assert _pt(cv, out, inf) == _pt(cv, want, winf)
    with pytest.raises(z.ZktError) as e:
        ctx.msm(np.zeros((301, 4), dtype=np.uint64))
    assert e.value.code == 5


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [10, 14, 16, 17, 18, 19])   # every digit-width / launch regime of msm.hip below 2^20
def test_msm_random_matches_oracle(cv, log_n, ctxs):
    ctx = ctxs[cv.name]
    n = (1 << log_n) + 5
    ctx.srs_generate(0xC0FFEE + log_n, n)
    srs = ctx.srs_download(0, n)
    rng = np.random.default_rng(log_n)
    sc = rand_fr(rng, n)
    sc[::97] = 0                                  # ~1 % zeros
    sc[1::89] = K.fr_to_mont(cv, [1])[0]          # ~1 % ones
    out, inf = ctx.msm(sc)
    want, winf = K.msm_mont(cv, srs, sc)
    assert not inf and np.array_equal(out, want)


@pytest.mark.parametrize("log_n", [16, 17])
@pytest.mark.parametrize("shape", ["all_equal", "small", "three_values", "sparse", "top_bits"])
def test_msm_skewed_digit_distributions(shape, log_n, ctxs):
    """The bucket grouping (two-level counting sort) and the chunked accumulation must not depend on the digits
    being uniform: a single crowded bucket per window (many level-2 tiles in one bin, the heavy-bucket fold),
    empty high windows, a handful of distinct digits, mostly-zero scalars."""
    cv = F.BN254
    ctx = ctxs[cv.name]
    n = (1 << log_n) + 321                        # 2^17: fifteen 17-bit windows, inlined reduction (r05 rules)
    ctx.srs_generate(0xD15EA5E, n)
    srs = ctx.srs_download(0, n)
    rng = np.random.default_rng(sum(map(ord, shape)))
    p = cv.fr.p
    if shape == "all_equal":
        vals = [0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF12345678 % p] * n
    elif shape == "small":
        vals = [int(x) for x in rng.integers(0, 1 << 16, size=n)]
    elif shape == "three_values":
        pool = [p - 1, (1 << 253) % p, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF]
        vals = [pool[int(i)] for i in rng.integers(0, 3, size=n)]
    elif shape == "sparse":
        vals = [0] * n
        for i in rng.integers(0, n, size=50):
            vals[int(i)] = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % p
    else:
        vals = [(int(x) << 238) % p for x in rng.integers(1, 1 << 15, size=n)]
    sc = K.fr_to_mont(cv, vals)
    out, inf = ctx.msm(sc)
    want, winf = K.msm_mont(cv, srs, sc)
    assert inf == winf and np.array_equal(out, want)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("shards", [2, 3, 8])
def test_msm_sharded_by_index_range(cv, shards, ctxs):
    """SURVEY.md section 8e "MSM - one tiny exchange": the partial sums over contiguous index ranges (what each GPU
    of a node would compute over its SRS slice), combined by the host-side point sum that follows the all-gather,
    equal the single-GPU MSM and the oracle."""
    from zkt_plonk_amd._lib import g1_sum_host
    from zkt_plonk_amd.parallel import shard_range
    ctx = ctxs[cv.name]
    n = 5000
    ctx.srs_generate(0x8E + shards, n)
    srs = ctx.srs_download(0, n)
    rng = np.random.default_rng(shards)
    sc = rand_fr(rng, n)
    full, finf = ctx.msm(sc)
    parts = []
    for r in range(shards):
        lo, hi = shard_range(n, r, shards)
        xy, inf = ctx.msm(sc[lo:hi], base_offset=lo)
        parts.append(np.zeros_like(xy) if inf else xy)
    got, ginf = g1_sum_host(cv.name, np.stack(parts))
    want, winf = K.msm_mont(cv, srs, sc)
    assert ginf == finf == winf and np.array_equal(got, full) and np.array_equal(got, want)


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_msm_full_size_2_20(cv, ctxs):
    """BASELINE config 2: 2^20 scalars/points, bit-exact commitment on both curves, plus linearity."""
    ctx = ctxs[cv.name]
    n = 1 << 20
    ctx.srs_generate(0x5EED, n)
    # the digit width the setup picks by itself: 17 on BN254, 19 on BLS12-381 (its additions cost 2.4 x as much: fewer, wider)
    assert ctx.msm_info()["window_bits"] == (17 if cv.name == "bn254" else 19) or os.environ.get("ZKT_MSM_CBITS")
    srs = ctx.srs_download(0, n)
    rng = np.random.default_rng(2020)
    a = rand_fr(rng, n)
    out, inf = ctx.msm(a)
    want, winf = K.msm_mont(cv, srs, a)
    assert not inf and np.array_equal(out, want)
    # linearity: msm(a) + msm(b) == msm(a + b) with b = a shifted (checked with oracle point adds)
    b = np.roll(a, 1, axis=0)
    ab = K.fr_to_mont(cv, [(x + y) % cv.fr.p for x, y in zip(K.fr_from_mont(cv, a[:4096]), K.fr_from_mont(cv, b[:4096]))])
    pa = _pt(cv, *ctx.msm(a[:4096]))
    pb = _pt(cv, *ctx.msm(b[:4096]))
    pab = _pt(cv, *ctx.msm(ab))
    assert C.add(cv, pa, pb) == pab


def test_msm_2_21_wide_pairs_and_wide_digits(ctxs):
    """BN254 at n = 2^21: the table index no longer fits a 4-byte pair (26 index bits), so the level-1 split writes
    (key, value) pairs; from this size on the digits are 19 bits wide (14 windows, 2^18 buckets), which takes the
    1024-column level-2 tables (and DigitLayout<5>) -- a combination the prover's configs do not reach on this curve.
    Canonical and Montgomery scalars, against the oracle."""
    cv = F.BN254
    ctx = ctxs[cv.name]
    n = 1 << 21
    ctx.srs_generate(0xA5A5, n)
    assert (ctx.msm_info()["windows"], ctx.msm_info()["window_bits"]) == (14, 19) or os.environ.get("ZKT_MSM_CBITS")
    srs = ctx.srs_download(0, n)
    assert np.array_equal(srs[:32], K.srs_mont(cv, 0xA5A5, 32))          # the table's R^-1 scaling is undone on the way out
    rng = np.random.default_rng(21)
    a = rand_fr(rng, n)
    a[::1000] = 0
    out, inf = ctx.msm(a)
    want, winf = K.msm_mont(cv, srs, a)
    assert not inf and np.array_equal(out, want)
    canon = K.ints_to_limbs(K.fr_from_mont(cv, a[:50000]), 4)
    out2, inf2 = ctx.msm(canon, montgomery=False)
    want2, _ = K.msm_mont(cv, srs[:50000], a[:50000])
    assert np.array_equal(out2, want2)
    ctx.srs_generate(0x5EED, 64)                                            # release the 2 GiB table
