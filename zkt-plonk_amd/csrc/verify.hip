// Verifier, everything but the pairings: plonk-core/src/proof_system/proof.rs:285-503 up to the two
// SonicKZG10::check calls.  Host-only code: one proof is 13 + 9 + 4 short scalar multiplications (~4 ms of host time on
// the 64-bit-limb arithmetic of hostec.hpp); a GPU has nothing to add at that size.  The result is, for each of the
// two openings, the pair (L, W) with
//     L = sum_i eta^i C_i - (sum_i eta^i v_i) G + z W
// so that the opening is valid iff e(L, h) == e(W, beta h) -- the two pairings stay with the caller (arkworks'
// `PairingEngine::product_of_pairings`), which also holds h and beta h (SonicKZG10's VerifierKey).
#include "ctx.hpp"
#include "ec.hpp"
#include "hostec.hpp"
#include "hostinv.hpp"
#include "pairing.hpp"
#include "transcript.hpp"

#include <cstring>
#include <memory>
#include <vector>

namespace zkt {

template <class P>
static Fe<P> pow_words(const Fe<P>& a, const uint32_t* e, int nwords) {
    Fe<P> r = fe_one<P>();
    bool started = false;
    for (int i = nwords * 32 - 1; i >= 0; --i) {
        if (started) r = fe_sqr<P>(r);
        if ((e[i / 32] >> (i % 32)) & 1u) {
            r = fe_mul<P>(r, a);
            started = true;
        }
    }
    return r;
}

template <class P>
static bool canonical_from_bytes(const uint8_t* b, uint32_t strip_top_bits, Fe<P>* out_mont, Fe<P>* out_canon = nullptr) {
    Fe<P> v;
    memcpy(v.v, b, P::N * 4);
    if (strip_top_bits) v.v[P::N - 1] &= (0xFFFFFFFFu >> strip_top_bits);
    for (int i = P::N - 1; i >= 0; --i) {
        if (v.v[i] < P::mod(i)) break;
        if (v.v[i] > P::mod(i) || i == 0) return false;
    }
    if (out_canon) *out_canon = v;
    *out_mont = fe_to_mont<P>(v);
    return true;
}

// a > -a as canonical integers (ark-serialize SWFlags: bit 7 = "y is the larger root")
template <class Q>
static bool is_larger_root(const Fe<Q>& y_mont) {
    const Fe<Q> y = fe_from_mont<Q>(y_mont), ny = fe_from_mont<Q>(fe_neg<Q>(y_mont));
    for (int i = Q::N - 1; i >= 0; --i)
        if (y.v[i] != ny.v[i]) return y.v[i] > ny.v[i];
    return false;
}

// a^e on 64-bit limbs, fixed 4-bit window (e: little-endian words)
template <class Q>
static hostec::HF<Q> hf_pow(const hostec::HF<Q>& a, const uint64_t* e, int nwords) {
    using hostec::HF;
    HF<Q> tab[16];
    tab[0] = hostec::hf_from<Q>(fe_one<Q>());
    for (int i = 1; i < 16; ++i) tab[i] = hostec::hf_mul<Q>(tab[i - 1], a);
    HF<Q> r = tab[0];
    bool started = false;
    for (int i = nwords * 16 - 1; i >= 0; --i) {
        const unsigned d = (unsigned)(e[i / 16] >> (4 * (i % 16))) & 15u;
        if (started)
            for (int k = 0; k < 4; ++k) r = hostec::hf_mul<Q>(r, r);
        if (d) {
            r = started ? hostec::hf_mul<Q>(r, tab[d]) : tab[d];
            started = true;
        }
    }
    return r;
}

// G1 = E(Fq)[r].  BN254 has cofactor one.  On BLS12-381 the curve has the endomorphism sigma(x, y) = (beta x, y) with
// sigma^2 + sigma + 1 = 0, acting on G1 as lambda = -x^2 (x the curve parameter; lambda^2 + lambda + 1 = r).  As
// endomorphisms (sigma - lambda)(sigma - lambda') = sigma^2 + sigma + 1 - r = -r for lambda' = x^2 - 1, so
// sigma(P) = [lambda] P implies [r] P = 0: two multiplications by the 64-bit |x| replace the 255-bit one of ark-ec's
// is_in_correct_subgroup_assuming_on_curve, with the same answer on every point of the curve.
template <class C>
static bool g1_in_subgroup(const Affine<typename C::Fq>& p) {
    using Q = typename C::Fq;
    if (C::ID == 0 || aff_is_inf<Q>(p)) return true;
    using namespace hostec;
    // beta = 2^((p - 1) / 3): the cube root of unity whose sigma is [-x^2] on G1 (tests/test_verify_host.py pins it)
    static const HF<Q> beta = [] {
        uint64_t e[HF<Q>::N], rem = 0;
        for (int i = HF<Q>::N - 1; i >= 0; --i) {
            const u128 cur = ((u128)rem << 64) | (HParams<Q>::mod(i) - (i == 0 ? 1 : 0));
            e[i] = (uint64_t)(cur / 3);
            rem = (uint64_t)(cur % 3);
        }
        return hf_pow<Q>(hf_from<Q>(fe_from_u32<Q>(2)), e, HF<Q>::N);
    }();
    const uint64_t xabs = 0xd201000000010000ULL;
    auto mul_x = [&](const HX<Q>& a) {
        HX<Q> acc = a;
        for (int i = 62; i >= 0; --i) {
            acc = hx_double<Q>(acc);
            if ((xabs >> i) & 1) acc = hx_add<Q>(acc, a);
        }
        return acc;
    };
    const HX<Q> P = hx_from<Q>(xyzz_from_affine<Q>(p));
    const HX<Q> x2p = mul_x(mul_x(P));                       // [x^2] P
    if (hf_is_zero<Q>(x2p.zz)) return false;                 // sigma(P) is never the identity
    // sigma(P) == -[x^2] P, compared projectively: X = beta x ZZ, Y = -y ZZZ
    const HF<Q> lx = hf_mul<Q>(hf_mul<Q>(beta, hf_from<Q>(p.x)), x2p.zz);
    const HF<Q> ly = hf_mul<Q>(hf_from<Q>(fe_neg<Q>(p.y)), x2p.zzz);
    return memcmp(lx.v, x2p.x.v, sizeof(lx.v)) == 0 && memcmp(ly.v, x2p.y.v, sizeof(ly.v)) == 0;
}

// compressed short-Weierstrass point of ark-serialize 0.3 (proof.rs:98-155 wire format) -> affine (Montgomery);
// both base fields have p = 3 mod 4, so sqrt(a) = a^((p + 1) / 4).  Checked deserialisation, as the reference's
// verifier relies on it (proof.rs:308 "subgroup checks are done when the proof is deserialised"): SWFlags::from_u8
// refuses both flag bits at once, x must be a canonical field element, the point must lie on the curve and in the
// prime-order subgroup.  Under the infinity flag the x bytes are parsed like any field element (so they must be below the
// modulus) and then ignored: GroupAffine::deserialize of ark-ec 0.3 reads (x, flags) with deserialize_with_flags and returns
// zero() when flags.is_infinity() -- as recalled, the crate is not in this image ("parity unpinned"; r03 refused a non-zero x
// there, which was stricter than the reference).
template <class C>
static bool decompress(const uint8_t* b, Affine<typename C::Fq>* out, bool* is_inf) {
    using Q = typename C::Fq;
    using hostec::HF;
    const size_t nb = Q::N * 4;
    const uint8_t flags = b[nb - 1];
    if ((flags & 0xC0) == 0xC0) return false;
    if (flags & 0x40) {
        Fe<Q> ignored;
        if (!canonical_from_bytes<Q>(b, 2, &ignored)) return false;
        *is_inf = true;
        out->x = fe_zero<Q>();
        out->y = fe_zero<Q>();
        return true;
    }
    *is_inf = false;
    Fe<Q> x;
    if (!canonical_from_bytes<Q>(b, 2, &x)) return false;
    const HF<Q> hx = hostec::hf_from<Q>(x);
    const HF<Q> rhs = hostec::hf_add<Q>(hostec::hf_mul<Q>(hostec::hf_mul<Q>(hx, hx), hx), hostec::hf_from<Q>(fe_from_u32<Q>(C::B)));
    static const std::vector<uint64_t> e = [] {   // (p + 1) / 4
        std::vector<uint64_t> w(HF<Q>::N + 1, 0);
        hostec::u128 carry = 1;
        for (int i = 0; i < HF<Q>::N; ++i) {
            carry += hostec::HParams<Q>::mod(i);
            w[i] = (uint64_t)carry;
            carry >>= 64;
        }
        w[HF<Q>::N] = (uint64_t)carry;
        for (int i = 0; i < HF<Q>::N; ++i) w[i] = (w[i] >> 2) | (w[i + 1] << 62);
        w.pop_back();
        return w;
    }();
    const HF<Q> hy = hf_pow<Q>(rhs, e.data(), HF<Q>::N);
    if (memcmp(hostec::hf_mul<Q>(hy, hy).v, rhs.v, sizeof(rhs.v)) != 0) return false;   // x is not on the curve
    Fe<Q> y = hostec::hf_to<Q>(hy);
    if (is_larger_root<Q>(y) != ((flags & 0x80) != 0)) y = fe_neg<Q>(y);
    out->x = x;
    out->y = y;
    return g1_in_subgroup<C>(*out);
}

// sum_i s_i P_i for a handful of points: interleaved width-5 NAFs over one doubling chain (the verifier's ~25-term
// combinations; double-and-add per term cost four times as much)
template <class C>
static hostec::HX<typename C::Fq> msm_wnaf(const hostec::HX<typename C::Fq>* pts, const Fe<typename C::Fr>* scalars_mont, int n) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    using namespace hostec;
    constexpr int BITS = R::N * 32 + 1;
    std::vector<int8_t> naf((size_t)n * BITS, 0);
    std::vector<HX<Q>> tab((size_t)n * 8);
    int top = 0;
    for (int t = 0; t < n; ++t) {
        Fe<R> s = fe_from_mont<R>(scalars_mont[t]);
        uint32_t k[R::N + 1];
        for (int i = 0; i < R::N; ++i) k[i] = s.v[i];
        k[R::N] = 0;
        auto is_zero = [&] { uint32_t o = 0; for (int i = 0; i <= R::N; ++i) o |= k[i]; return o == 0; };
        int pos = 0;
        while (!is_zero()) {
            int d = 0;
            if (k[0] & 1u) {
                d = (int)(k[0] & 31u);
                if (d >= 16) d -= 32;
                // k -= d
                int64_t carry = -(int64_t)d;
                for (int i = 0; i <= R::N && carry; ++i) {
                    const int64_t v = (int64_t)k[i] + carry;
                    k[i] = (uint32_t)v;
                    carry = v >> 32;
                }
            }
            naf[(size_t)t * BITS + pos] = (int8_t)d;
            for (int i = 0; i < R::N; ++i) k[i] = (k[i] >> 1) | (k[i + 1] << 31);
            k[R::N] >>= 1;
            ++pos;
        }
        if (pos > top) top = pos;
        // odd multiples P, 3P, ..., 15P
        const HX<Q> p2 = hx_double<Q>(pts[t]);
        tab[(size_t)t * 8] = pts[t];
        for (int j = 1; j < 8; ++j) tab[(size_t)t * 8 + j] = hx_add<Q>(tab[(size_t)t * 8 + j - 1], p2);
    }
    HX<Q> acc = hx_identity<Q>();
    for (int i = top - 1; i >= 0; --i) {
        acc = hx_double<Q>(acc);
        for (int t = 0; t < n; ++t) {
            const int d = naf[(size_t)t * BITS + i];
            if (!d) continue;
            HX<Q> e = tab[(size_t)t * 8 + (size_t)((d > 0 ? d : -d) >> 1)];
            if (d < 0) e.y = hf_sub<Q>(HF<Q>{}, e.y);
            acc = hx_add<Q>(acc, e);
        }
    }
    return acc;
}

template <class C>
struct Verifier {
    using R = typename C::Fr;
    using Q = typename C::Fq;
    using F = Fe<R>;
    using HX = hostec::HX<Q>;

    static HX to_hx(const Affine<Q>& a) { return hostec::hx_from<Q>(xyzz_from_affine<Q>(a)); }
    static HX mul(const HX& p, const F& s_mont) {   // double-and-add on the canonical scalar
        const F s = fe_from_mont<R>(s_mont);
        HX acc = hostec::hx_identity<Q>();
        bool started = false;
        for (int i = R::N * 32 - 1; i >= 0; --i) {
            if (started) acc = hostec::hx_double<Q>(acc);
            if ((s.v[i / 32] >> (i % 32)) & 1u) {
                acc = hostec::hx_add<Q>(acc, p);
                started = true;
            }
        }
        return acc;
    }
    static void to_le(const F& mont, uint8_t out[32]) {
        const F c = fe_from_mont<R>(mont);
        memcpy(out, c.v, 32);
    }
    static F from_le(const uint8_t in[32]) {
        F c;
        memcpy(c.v, in, 32);
        return fe_to_mont<R>(c);
    }
    static void tr_commit(HostTranscript& tr, const char* label, const Affine<Q>& p) {
        uint8_t x[64] = {0}, y[64] = {0};
        const bool inf = aff_is_inf<Q>(p);
        if (!inf) {
            const Fe<Q> cx = fe_from_mont<Q>(p.x), cy = fe_from_mont<Q>(p.y);
            memcpy(x, cx.v, Q::N * 4);
            memcpy(y, cy.v, Q::N * 4);
        }
        tr.append_commitment(label, x, y, Q::N * 4, inf);
    }
    static F challenge(HostTranscript& tr, const char* label) {
        uint8_t b[32];
        tr.challenge_scalar(label, R::BITS, b);
        return from_le(b);
    }

    static int run(const zkt_verify_inputs& in, HostTranscript& tr, uint64_t* out_pairs, int* out_inf) {
        const size_t nb = Q::N * 4;
        const int L64 = Q::N / 2;
        if (in.proof_len != 13 * nb + 2 + 12 * 32) return ZKT_ERR_INVALID_ARGUMENT;
        if (in.n == 0 || (in.n & (in.n - 1))) return ZKT_ERR_INVALID_DOMAIN_SIZE;
        int log_n = 0;
        while (((uint64_t)1 << log_n) < in.n) ++log_n;
        if (log_n > R::TWO_ADICITY) return ZKT_ERR_INVALID_DOMAIN_SIZE;
        // ---- Proof::deserialize (proof.rs:98-155): 11 commitments, 2 x (opening, Option::None), 12 evaluations ----
        Affine<Q> cm[13];
        size_t pos = 0;
        for (int k = 0; k < 13; ++k) {
            bool inf = false;
            if (!decompress<C>(in.proof + pos, &cm[k], &inf)) return ZKT_ERR_INVALID_ARGUMENT;
            pos += nb;
            if (k >= 11) {
                if (in.proof[pos] != 0) return ZKT_ERR_INVALID_ARGUMENT;   // kzg10::Proof::random_v must be None
                pos += 1;
            }
        }
        F ev[12];
        for (int k = 0; k < 12; ++k) {
            if (!canonical_from_bytes<R>(in.proof + pos, 0, &ev[k])) return ZKT_ERR_INVALID_ARGUMENT;
            pos += 32;
        }
        enum { A, B, Cc, T, H1, H2, Z1, Z2, QLO, QMID, QHI, AW, SAW };
        const F &e_a = ev[0], &e_b = ev[1], &e_c = ev[2], &e_s1 = ev[3], &e_s2 = ev[4], &e_z1n = ev[5], &e_ql = ev[6],
                &e_t = ev[7], &e_tn = ev[8], &e_z2n = ev[9], &e_h1n = ev[10], &e_h2 = ev[11];
        Affine<Q> vk[10];
        for (int k = 0; k < 10; ++k) {
            if (in.vk_is_infinity && in.vk_is_infinity[k]) {
                vk[k].x = fe_zero<Q>();
                vk[k].y = fe_zero<Q>();
            } else {
                memcpy(vk[k].x.v, in.vk_commitments + (size_t)k * 2 * L64, nb);
                memcpy(vk[k].y.v, in.vk_commitments + (size_t)k * 2 * L64 + L64, nb);
            }
        }
        enum { QM, QL, QR, QO, QC, S1, S2, S3, QLOOKUP, QTABLE };
        // ---- transcript (proof.rs:308-360) ----
        {
            std::vector<uint8_t> b(in.n_pi * 32);
            for (size_t i = 0; i < in.n_pi; ++i) {
                F v;
                memcpy(v.v, in.pub_inputs + 4 * i, 32);
                to_le(v, b.data() + 32 * i);
            }
            tr.append_scalars("pi", b.data(), in.n_pi, 32, false);
        }
        tr_commit(tr, "a_commit", cm[A]); tr_commit(tr, "b_commit", cm[B]); tr_commit(tr, "c_commit", cm[Cc]);
        tr_commit(tr, "t_commit", cm[T]); tr_commit(tr, "h1_commit", cm[H1]); tr_commit(tr, "h2_commit", cm[H2]);
        const F beta = challenge(tr, "beta"), gamma = challenge(tr, "gamma"), delta = challenge(tr, "delta"),
                epsilon = challenge(tr, "epsilon");
        if (fe_eq<R>(beta, gamma) || fe_eq<R>(beta, delta) || fe_eq<R>(beta, epsilon) || fe_eq<R>(gamma, delta) ||
            fe_eq<R>(gamma, epsilon) || fe_eq<R>(delta, epsilon))
            return ZKT_ERR_EQUAL_CHALLENGES;
        tr_commit(tr, "z1_commit", cm[Z1]); tr_commit(tr, "z2_commit", cm[Z2]);
        const F alpha = challenge(tr, "alpha");
        tr_commit(tr, "q_lo_commit", cm[QLO]); tr_commit(tr, "q_mid_commit", cm[QMID]); tr_commit(tr, "q_hi_commit", cm[QHI]);
        const F xi = challenge(tr, "xi");
        // ---- scalars (proof.rs:362-378, compute_r0 163-217, util.rs:185-195) ----
        const F one = fe_one<R>();
        const F zh = fe_sub<R>(fe_pow_u64<R>(xi, in.n), one);
        F nn = fe_zero<R>();
        nn.v[0] = (uint32_t)(in.n & 0xffffffffu);
        nn.v[1] = (uint32_t)(in.n >> 32);
        nn = fe_to_mont<R>(nn);
        auto lagrange = [&](const F& point, F* out) {   // zh * point / (n * (xi - point))
            const F den = fe_mul<R>(nn, fe_sub<R>(xi, point));
            if (fe_is_zero<R>(den)) return false;
            *out = fe_mul<R>(fe_mul<R>(zh, point), fe_inv_host<R>(den));
            return true;
        };
        F l1;
        if (!lagrange(one, &l1)) return ZKT_ERR_ZERO_DENOMINATOR;
        const F a2 = fe_sqr<R>(alpha), a3 = fe_mul<R>(a2, alpha), a4 = fe_mul<R>(a3, alpha), a5 = fe_mul<R>(a4, alpha);
        const F opd = fe_add<R>(one, delta), eopd = fe_mul<R>(epsilon, opd);
        const F k1 = fe_from_u32<R>(7), k2 = fe_from_u32<R>(13);
        F r0 = fe_zero<R>();
        for (size_t i = 0; i < in.n_pi; ++i) {
            F root, v, li;
            memcpy(root.v, in.pi_roots + 4 * i, 32);
            memcpy(v.v, in.pub_inputs + 4 * i, 32);
            if (!lagrange(root, &li)) return ZKT_ERR_ZERO_DENOMINATOR;
            r0 = fe_sub<R>(r0, fe_mul<R>(li, v));
        }
        {
            F p2 = fe_mul<R>(alpha, e_z1n);
            p2 = fe_mul<R>(p2, fe_add<R>(fe_add<R>(e_a, fe_mul<R>(beta, e_s1)), gamma));
            p2 = fe_mul<R>(p2, fe_add<R>(fe_add<R>(e_b, fe_mul<R>(beta, e_s2)), gamma));
            p2 = fe_mul<R>(p2, fe_add<R>(e_c, gamma));
            F p4 = fe_mul<R>(a3, e_z2n);
            p4 = fe_mul<R>(p4, fe_add<R>(eopd, fe_mul<R>(delta, e_h2)));
            p4 = fe_mul<R>(p4, fe_add<R>(fe_add<R>(eopd, e_h2), fe_mul<R>(delta, e_h1n)));
            r0 = fe_add<R>(fe_add<R>(r0, p2), fe_add<R>(fe_mul<R>(l1, a2), fe_add<R>(p4, fe_mul<R>(l1, a4))));
        }
        // ---- linearisation commitment (proof.rs:220-282; the 13-point MSM of commitment.rs:32-45) ----
        const F bz = fe_mul<R>(beta, xi);
        F s_z1 = fe_mul<R>(alpha, fe_add<R>(fe_add<R>(bz, e_a), gamma));
        s_z1 = fe_mul<R>(s_z1, fe_add<R>(fe_add<R>(fe_mul<R>(bz, k1), e_b), gamma));
        s_z1 = fe_mul<R>(s_z1, fe_add<R>(fe_add<R>(fe_mul<R>(bz, k2), e_c), gamma));
        s_z1 = fe_add<R>(s_z1, fe_mul<R>(l1, a2));
        F s_s3 = fe_mul<R>(fe_mul<R>(fe_neg<R>(alpha), beta), e_z1n);
        s_s3 = fe_mul<R>(s_s3, fe_add<R>(fe_add<R>(fe_mul<R>(beta, e_s1), e_a), gamma));
        s_s3 = fe_mul<R>(s_s3, fe_add<R>(fe_add<R>(fe_mul<R>(beta, e_s2), e_b), gamma));
        F s_z2 = fe_mul<R>(fe_mul<R>(a3, opd), fe_add<R>(epsilon, fe_mul<R>(e_ql, e_c)));
        s_z2 = fe_mul<R>(s_z2, fe_add<R>(fe_add<R>(eopd, e_t), fe_mul<R>(delta, e_tn)));
        s_z2 = fe_add<R>(s_z2, fe_mul<R>(a4, l1));
        F s_h1 = fe_mul<R>(fe_mul<R>(fe_neg<R>(a3), e_z2n), fe_add<R>(fe_add<R>(eopd, e_h2), fe_mul<R>(delta, e_h1n)));
        const F s_qt = fe_mul<R>(a5, e_t);
        const F xn2 = fe_mul<R>(fe_mul<R>(fe_add<R>(zh, one), xi), xi), nzh = fe_neg<R>(zh);
        const F sc[13] = {fe_mul<R>(e_a, e_b), e_a, e_b, e_c, one, s_z1, s_s3, s_z2, s_h1, s_qt,
                          nzh, fe_mul<R>(nzh, xn2), fe_mul<R>(nzh, fe_sqr<R>(xn2))};
        const Affine<Q>* pt[13] = {&vk[QM], &vk[QL], &vk[QR], &vk[QO], &vk[QC], &cm[Z1], &vk[S3], &cm[Z2], &cm[H1], &vk[QTABLE],
                                   &cm[QLO], &cm[QMID], &cm[QHI]};
        static const char* EL[12] = {"a_eval", "b_eval", "c_eval", "sigma1_eval", "sigma2_eval", "z1_next_eval",
                                     "q_lookup_eval", "t_eval", "t_next_eval", "z2_next_eval", "h1_next_eval", "h2_eval"};
        for (int k = 0; k < 12; ++k) {
            uint8_t b[32];
            to_le(ev[k], b);
            tr.append_scalars(EL[k], b, 1, 32, true);
        }
        const F eta = challenge(tr, "eta");
        // ---- the two openings (proof.rs:420-500) ----
        // L = sum_i eta^i C_i - (sum_i eta^i v_i) g + z W as ONE short multi-scalar multiplication; the linearisation
        // commitment (C_0 of the first opening, coefficient eta^0 = 1) enters through its own 13 terms
        Affine<Q> g;
        memcpy(g.x.v, in.g, nb);
        memcpy(g.y.v, in.g + L64, nb);
        const F w = root_of_unity<R>(log_n);
        auto emit = [&](std::vector<HX>& pts, std::vector<F>& scs, const Affine<Q>& wit, int slot) {
            const HX L = msm_wnaf<C>(pts.data(), scs.data(), (int)pts.size());
            const Affine<Q> la = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(L));
            uint64_t* o = out_pairs + (size_t)slot * 4 * L64;
            memcpy(o, la.x.v, nb);
            memcpy(o + L64, la.y.v, nb);
            memcpy(o + 2 * L64, wit.x.v, nb);
            memcpy(o + 3 * L64, wit.y.v, nb);
            if (out_inf) {
                out_inf[2 * slot] = aff_is_inf<Q>(la) ? 1 : 0;
                out_inf[2 * slot + 1] = aff_is_inf<Q>(wit) ? 1 : 0;
            }
        };
        auto fold = [&](std::vector<HX>& pts, std::vector<F>& scs, const Affine<Q>* const* commits, const F* values, int k,
                        F ch, F comb_v, const F& z, const Affine<Q>& wit) {
            for (int i = 0; i < k; ++i) {
                pts.push_back(to_hx(*commits[i]));
                scs.push_back(ch);
                comb_v = fe_add<R>(comb_v, fe_mul<R>(ch, values[i]));
                ch = fe_mul<R>(ch, eta);
            }
            pts.push_back(to_hx(g));
            scs.push_back(fe_neg<R>(comb_v));
            pts.push_back(to_hx(wit));
            scs.push_back(z);
        };
        {
            std::vector<HX> pts;
            std::vector<F> scs;
            for (int k = 0; k < 13; ++k) {
                pts.push_back(to_hx(*pt[k]));
                scs.push_back(sc[k]);
            }
            const Affine<Q>* cs[8] = {&cm[A], &cm[B], &cm[Cc], &vk[S1], &vk[S2], &vk[QLOOKUP], &cm[T], &cm[H2]};
            const F vs[8] = {e_a, e_b, e_c, e_s1, e_s2, e_ql, e_t, e_h2};
            fold(pts, scs, cs, vs, 8, eta, r0, xi, cm[AW]);
            emit(pts, scs, cm[AW], 0);
        }
        {
            std::vector<HX> pts;
            std::vector<F> scs;
            const Affine<Q>* cs[4] = {&cm[Z1], &cm[Z2], &cm[T], &cm[H1]};
            const F vs[4] = {e_z1n, e_z2n, e_tn, e_h1n};
            fold(pts, scs, cs, vs, 4, one, fe_zero<R>(), fe_mul<R>(xi, w), cm[SAW]);
            emit(pts, scs, cm[SAW], 1);
        }
        return ZKT_OK;
    }
};

}  // namespace zkt

using namespace zkt;

template <class C>
static void g1_msm_host_t(const uint64_t* pts, const uint64_t* scalars, size_t n, int scalars_mont, uint64_t* out, int* out_inf) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    constexpr int L64 = Q::N / 2;
    typename Verifier<C>::HX acc = hostec::hx_identity<Q>();
    for (size_t i = 0; i < n; ++i) {
        Affine<Q> a;
        memcpy(a.x.v, pts + i * 2 * L64, Q::N * 4);
        memcpy(a.y.v, pts + i * 2 * L64 + L64, Q::N * 4);
        if (aff_is_inf<Q>(a)) continue;
        Fe<R> s;
        memcpy(s.v, scalars + 4 * i, 32);
        if (!scalars_mont) s = fe_to_mont<R>(s);
        acc = hostec::hx_add<Q>(acc, Verifier<C>::mul(Verifier<C>::to_hx(a), s));
    }
    const Affine<Q> r = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(acc));
    memcpy(out, r.x.v, Q::N * 4);
    memcpy(out + L64, r.y.v, Q::N * 4);
    if (out_inf) *out_inf = aff_is_inf<Q>(r) ? 1 : 0;
}

extern "C" int zkt_g1_msm_host(int curve_id, const uint64_t* points_xy_mont, const uint64_t* scalars, size_t n,
                               int scalars_montgomery, uint64_t* out_xy_mont, int* out_is_infinity) {
    if ((n && (!points_xy_mont || !scalars)) || !out_xy_mont) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) g1_msm_host_t<Bn254Curve>(points_xy_mont, scalars, n, scalars_montgomery, out_xy_mont, out_is_infinity);
    else if (curve_id == ZKT_CURVE_BLS12_381) g1_msm_host_t<Bls381Curve>(points_xy_mont, scalars, n, scalars_montgomery, out_xy_mont, out_is_infinity);
    else return ZKT_ERR_INVALID_ARGUMENT;
    return ZKT_OK;
}

template <class C>
static int pairing_inputs(const uint64_t* g1, const uint64_t* g2, size_t n, std::vector<typename pairing::Tower<C>::G1>& ps,
                          std::vector<typename pairing::Tower<C>::G2>& qs) {
    using Q = typename C::Fq;
    using T = pairing::Tower<C>;
    constexpr int L64 = Q::N / 2;
    ps.resize(n);
    qs.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Fe<Q> c[6];
        memcpy(c[0].v, g1 + i * 2 * L64, Q::N * 4);
        memcpy(c[1].v, g1 + i * 2 * L64 + L64, Q::N * 4);
        for (int k = 0; k < 4; ++k) memcpy(c[2 + k].v, g2 + i * 4 * L64 + (size_t)k * L64, Q::N * 4);
        ps[i].inf = fe_is_zero<Q>(c[0]) && fe_is_zero<Q>(c[1]);
        ps[i].x = hostec::hf_from<Q>(c[0]);
        ps[i].y = hostec::hf_from<Q>(c[1]);
        qs[i].inf = fe_is_zero<Q>(c[2]) && fe_is_zero<Q>(c[3]) && fe_is_zero<Q>(c[4]) && fe_is_zero<Q>(c[5]);
        qs[i].x = typename T::E2{hostec::hf_from<Q>(c[2]), hostec::hf_from<Q>(c[3])};
        qs[i].y = typename T::E2{hostec::hf_from<Q>(c[4]), hostec::hf_from<Q>(c[5])};
        if (!T::g1_on_curve(ps[i]) || !T::g2_on_twist(qs[i])) return ZKT_ERR_INVALID_ARGUMENT;
        // the pairing is bilinear on G1 x G2 only: a point of the curve outside the prime-order subgroup is refused
        Affine<Q> pa;
        pa.x = c[0];
        pa.y = c[1];
        if (!ps[i].inf && !g1_in_subgroup<C>(pa)) return ZKT_ERR_INVALID_ARGUMENT;
    }
    return ZKT_OK;
}

template <class C>
static int pairing_check_t(const uint64_t* g1, const uint64_t* g2, size_t n, int* is_one) {
    std::vector<typename pairing::Tower<C>::G1> ps;
    std::vector<typename pairing::Tower<C>::G2> qs;
    int rc = pairing_inputs<C>(g1, g2, n, ps, qs);
    if (rc) return rc;
    bool valid = true;
    *is_one = pairing::Tower<C>::product_is_one(ps.data(), qs.data(), n, &valid) ? 1 : 0;
    return valid ? ZKT_OK : ZKT_ERR_INVALID_ARGUMENT;   // a G2 argument of small order
}

// every shortcut of csrc/pairing.hpp against its plain definition; 0 = all good, else the number of the failing check
extern "C" int zkt_debug_pairing_selftest(int curve_id) {
    if (curve_id == ZKT_CURVE_BN254) return pairing::Tower<Bn254Curve>::selftest();
    if (curve_id == ZKT_CURVE_BLS12_381) return pairing::Tower<Bls381Curve>::selftest();
    return -1;
}

extern "C" int zkt_pairing_product_is_one(int curve_id, const uint64_t* g1_xy_mont, const uint64_t* g2_xy_mont, size_t n,
                                          int* is_one) {
    if (!is_one || (n && (!g1_xy_mont || !g2_xy_mont))) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return pairing_check_t<Bn254Curve>(g1_xy_mont, g2_xy_mont, n, is_one);
    if (curve_id == ZKT_CURVE_BLS12_381) return pairing_check_t<Bls381Curve>(g1_xy_mont, g2_xy_mont, n, is_one);
    return ZKT_ERR_INVALID_ARGUMENT;
}

extern "C" int zkt_verify_prepare(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, uint64_t* out_pairs,
                                  int* out_is_infinity);

// The whole of Proof::verify (proof.rs:285-503): zkt_verify_prepare, then the two SonicKZG10::check calls
// (proof.rs:441,479), e(L_k, h) e(-W_k, beta h) == 1 for k = 1, 2, folded into ONE product of two pairings the way
// ark-poly-commit's batch_check folds openings: with rho = the low 128 bits of Keccak-256 over (L1, W1, L2, W2, h, beta h),
//     e(L1 + rho L2, h) e(-(W1 + rho W2), beta h) == 1.
// Both checks hold => the folded one holds; if one of them fails, the folded one holds for at most one value of rho
// (the exponent is linear in it), i.e. with probability 2^-128 over the hash: the accept set is the reference's.
template <class C>
static int verify_t(int curve_id, const zkt_verify_inputs* in, zkt_transcript* tr, const uint64_t* h, const uint64_t* beta_h,
                    int* accepted) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    using HX = hostec::HX<Q>;
    constexpr int L64 = Q::N / 2;
    uint64_t pairs[4 * 12];
    int inf[4];
    int rc = zkt_verify_prepare(curve_id, in, tr, pairs, inf);
    if (rc) return rc;
    Fe<R> rho = fe_zero<R>();
    {
        std::vector<uint8_t> buf(4 + (size_t)(8 * L64 + 8 * L64) * 8);
        const uint32_t cid = (uint32_t)curve_id;
        memcpy(buf.data(), &cid, 4);
        memcpy(buf.data() + 4, pairs, (size_t)8 * L64 * 8);
        memcpy(buf.data() + 4 + (size_t)8 * L64 * 8, h, (size_t)4 * L64 * 8);
        memcpy(buf.data() + 4 + (size_t)12 * L64 * 8, beta_h, (size_t)4 * L64 * 8);
        uint8_t dg[32];
        keccak256(buf.data(), buf.size(), dg);
        memcpy(rho.v, dg, 16);                      // 128 bits: below both scalar moduli
        rho = fe_to_mont<R>(rho);
    }
    auto point = [&](int idx) {
        Affine<Q> a;
        memcpy(a.x.v, pairs + (size_t)idx * 2 * L64, L64 * 8);
        memcpy(a.y.v, pairs + (size_t)idx * 2 * L64 + L64, L64 * 8);
        return hostec::hx_from<Q>(xyzz_from_affine<Q>(a));
    };
    const Fe<R> sc[2] = {fe_one<R>(), rho};
    const HX lp[2] = {point(0), point(2)}, wp[2] = {point(1), point(3)};
    const Affine<Q> lc = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(msm_wnaf<C>(lp, sc, 2)));
    Affine<Q> wc = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(msm_wnaf<C>(wp, sc, 2)));
    if (!aff_is_inf<Q>(wc)) wc.y = fe_neg<Q>(wc.y);
    uint64_t g1[2 * 12], g2[2 * 4 * 6];
    memcpy(g1, lc.x.v, L64 * 8);
    memcpy(g1 + L64, lc.y.v, L64 * 8);
    memcpy(g1 + 2 * L64, wc.x.v, L64 * 8);
    memcpy(g1 + 3 * L64, wc.y.v, L64 * 8);
    memcpy(g2, h, 4 * L64 * 8);
    memcpy(g2 + 4 * L64, beta_h, 4 * L64 * 8);
    int one = 0;
    if ((rc = pairing_check_t<C>(g1, g2, 2, &one))) return rc;
    *accepted = one ? 1 : 0;    // 0: Error::ProofVerificationError
    return ZKT_OK;
}

// k proofs under the SAME structured reference string (h, beta h; the circuits and verifier keys may differ), ONE product of
// two pairings for all of them: every proof contributes its two folded openings (L_2i, W_2i), (L_2i+1, W_2i+1), and
//     e(sum_j rho_j L_j, h) e(-sum_j rho_j W_j, beta h) == 1,   rho_0 = 1, rho_j = low 128 bits of Keccak-256(seed || j),
// seed = Keccak-256 over the curve, k, every (L, W) and (h, beta h).  All 2k checks hold => the product is one; if any
// fails, the product is one for at most a 2^-128 fraction of the seeds.  A rejected batch says nothing about WHICH proof
// is bad: the caller falls back to zkt_verify per proof.  The reference has no such entry (its PC::check folds the two
// openings of one proof only, proof.rs:441,479); this is the service-side extension SURVEY.md 8f.4 asks for.
template <class C>
static int verify_batch_t(int curve_id, const zkt_verify_inputs* ins, zkt_transcript* const* trs, size_t k, const uint64_t* h,
                          const uint64_t* beta_h, int* accepted) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    using HX = hostec::HX<Q>;
    constexpr int L64 = Q::N / 2;
    const size_t pair_words = (size_t)8 * L64;               // L1, W1, L2, W2 of one proof
    std::vector<uint64_t> pairs(k * pair_words);
    for (size_t i = 0; i < k; ++i) {
        if (!trs[i]) return ZKT_ERR_INVALID_ARGUMENT;
        int inf[4];
        int rc = zkt_verify_prepare(curve_id, &ins[i], trs[i], pairs.data() + i * pair_words, inf);
        if (rc) return rc;
    }
    uint8_t seed[32];
    {
        std::vector<uint8_t> buf(4 + 8 + pairs.size() * 8 + (size_t)8 * L64 * 8);
        const uint32_t cid = (uint32_t)curve_id;
        const uint64_t kk = (uint64_t)k;
        size_t at = 0;
        memcpy(buf.data() + at, &cid, 4); at += 4;
        memcpy(buf.data() + at, &kk, 8); at += 8;
        memcpy(buf.data() + at, pairs.data(), pairs.size() * 8); at += pairs.size() * 8;
        memcpy(buf.data() + at, h, (size_t)4 * L64 * 8); at += (size_t)4 * L64 * 8;
        memcpy(buf.data() + at, beta_h, (size_t)4 * L64 * 8);
        keccak256(buf.data(), buf.size(), seed);
    }
    const size_t m = 2 * k;
    std::vector<Fe<R>> rho(m);
    std::vector<HX> lp(m), wp(m);
    for (size_t j = 0; j < m; ++j) {
        if (j == 0) {
            rho[j] = fe_one<R>();
        } else {
            uint8_t msg[40], dg[32];
            const uint64_t jj = (uint64_t)j;
            memcpy(msg, seed, 32);
            memcpy(msg + 32, &jj, 8);
            keccak256(msg, sizeof msg, dg);
            Fe<R> r = fe_zero<R>();
            memcpy(r.v, dg, 16);                       // 128 bits: below both scalar moduli
            rho[j] = fe_to_mont<R>(r);
        }
        auto point = [&](size_t idx) {
            Affine<Q> a;
            memcpy(a.x.v, pairs.data() + idx * 2 * L64, L64 * 8);
            memcpy(a.y.v, pairs.data() + idx * 2 * L64 + L64, L64 * 8);
            return hostec::hx_from<Q>(xyzz_from_affine<Q>(a));
        };
        lp[j] = point(2 * j);          // pairs are laid out L, W, L, W, ...
        wp[j] = point(2 * j + 1);
    }
    const Affine<Q> lc = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(msm_wnaf<C>(lp.data(), rho.data(), (int)m)));
    Affine<Q> wc = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(msm_wnaf<C>(wp.data(), rho.data(), (int)m)));
    if (!aff_is_inf<Q>(wc)) wc.y = fe_neg<Q>(wc.y);
    uint64_t g1[2 * 12], g2[2 * 4 * 6];
    memcpy(g1, lc.x.v, L64 * 8);
    memcpy(g1 + L64, lc.y.v, L64 * 8);
    memcpy(g1 + 2 * L64, wc.x.v, L64 * 8);
    memcpy(g1 + 3 * L64, wc.y.v, L64 * 8);
    memcpy(g2, h, 4 * L64 * 8);
    memcpy(g2 + 4 * L64, beta_h, 4 * L64 * 8);
    int one = 0;
    int rc = pairing_check_t<C>(g1, g2, 2, &one);
    if (rc) return rc;
    *accepted = one ? 1 : 0;
    return ZKT_OK;
}

// ---- the G2 half of the test / bench SRS (zkt_srs_generate is the G1 half): h = the G2 generator of ark-bn254 /
// ---- ark-bls12-381 (published constants), beta h = tau h by double-and-add on the twist (one-time, host) ----------
template <class C>
static int srs_g2_t(const uint64_t* tau4, uint64_t* out_h, uint64_t* out_beta_h) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    using T = pairing::Tower<C>;
    using E2 = typename T::E2;
    constexpr int L64 = Q::N / 2;
    static const uint64_t gen_bn[4][4] = {
        {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL},
        {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL},
        {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL},
        {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};
    static const uint64_t gen_bls[4][6] = {
        {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL},
        {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL},
        {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL},
        {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL}};
    hostec::HF<Q> c[4];
    for (int k = 0; k < 4; ++k) {
        Fe<Q> v;
        memcpy(v.v, C::ID == 0 ? (const void*)gen_bn[k] : (const void*)gen_bls[k], L64 * 8);
        c[k] = hostec::hf_from<Q>(fe_to_mont<Q>(v));
    }
    const E2 gx{c[0], c[1]}, gy{c[2], c[3]};
    typename T::G2 gq{gx, gy, false};
    if (!T::g2_on_twist(gq)) return ZKT_ERR_INVALID_ARGUMENT;
    Fe<R> tau;
    memcpy(tau.v, tau4, 32);
    // affine double-and-add; (x, y, inf)
    E2 ax = gx, ay = gy;
    bool ainf = true;
    auto add_pt = [&](const E2& px, const E2& py, bool pinf) {   // acc += p
        if (pinf) return;
        if (ainf) { ax = px; ay = py; ainf = false; return; }
        E2 lam, di;
        if (T::eq2(ax, px)) {
            if (!T::eq2(ay, py) || T::is_zero2(ay)) { ainf = true; return; }
            (void)T::inv2(T::dbl2(ay), &di);
            const E2 xx = T::sqr2(ax);
            lam = T::mul2(T::add2(T::dbl2(xx), xx), di);
        } else {
            (void)T::inv2(T::sub2(px, ax), &di);
            lam = T::mul2(T::sub2(py, ay), di);
        }
        const E2 nx = T::sub2(T::sub2(T::sqr2(lam), ax), px);
        ay = T::sub2(T::mul2(lam, T::sub2(ax, nx)), ay);
        ax = nx;
    };
    for (int i = R::N * 32 - 1; i >= 0; --i) {
        if (!ainf) add_pt(ax, ay, false);                         // double
        if ((tau.v[i / 32] >> (i % 32)) & 1u) add_pt(gx, gy, false);
    }
    memcpy(out_h, &gx, 2 * L64 * 8);
    memcpy(out_h + 2 * L64, &gy, 2 * L64 * 8);
    if (ainf) {
        memset(out_beta_h, 0, 4 * L64 * 8);
    } else {
        memcpy(out_beta_h, &ax, 2 * L64 * 8);
        memcpy(out_beta_h + 2 * L64, &ay, 2 * L64 * 8);
    }
    return ZKT_OK;
}

extern "C" int zkt_srs_generate_g2(int curve_id, const uint64_t* tau_canonical4, uint64_t* out_h, uint64_t* out_beta_h) {
    if (!tau_canonical4 || !out_h || !out_beta_h) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return srs_g2_t<Bn254Curve>(tau_canonical4, out_h, out_beta_h);
    if (curve_id == ZKT_CURVE_BLS12_381) return srs_g2_t<Bls381Curve>(tau_canonical4, out_h, out_beta_h);
    return ZKT_ERR_INVALID_ARGUMENT;
}

extern "C" int zkt_verify(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, const uint64_t* h_g2_mont,
                          const uint64_t* beta_h_g2_mont, int* accepted) {
    if (!in || !transcript || !h_g2_mont || !beta_h_g2_mont || !accepted) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return verify_t<Bn254Curve>(curve_id, in, transcript, h_g2_mont, beta_h_g2_mont, accepted);
    if (curve_id == ZKT_CURVE_BLS12_381) return verify_t<Bls381Curve>(curve_id, in, transcript, h_g2_mont, beta_h_g2_mont, accepted);
    return ZKT_ERR_INVALID_ARGUMENT;
}

extern "C" int zkt_verify_batch(int curve_id, const zkt_verify_inputs* ins, zkt_transcript* const* transcripts, size_t count,
                                const uint64_t* h_g2_mont, const uint64_t* beta_h_g2_mont, int* accepted) {
    if (!ins || !transcripts || count == 0 || count > (1u << 20) || !h_g2_mont || !beta_h_g2_mont || !accepted)
        return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254)
        return verify_batch_t<Bn254Curve>(curve_id, ins, transcripts, count, h_g2_mont, beta_h_g2_mont, accepted);
    if (curve_id == ZKT_CURVE_BLS12_381)
        return verify_batch_t<Bls381Curve>(curve_id, ins, transcripts, count, h_g2_mont, beta_h_g2_mont, accepted);
    return ZKT_ERR_INVALID_ARGUMENT;
}

extern "C" int zkt_verify_prepare(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, uint64_t* out_pairs,
                                  int* out_is_infinity) {
    if (!in || !transcript || !out_pairs || !in->proof || !in->vk_commitments || !in->g || (in->n_pi && (!in->pi_roots || !in->pub_inputs)))
        return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return Verifier<Bn254Curve>::run(*in, *transcript->impl, out_pairs, out_is_infinity);
    if (curve_id == ZKT_CURVE_BLS12_381) return Verifier<Bls381Curve>::run(*in, *transcript->impl, out_pairs, out_is_infinity);
    return ZKT_ERR_INVALID_ARGUMENT;
}
