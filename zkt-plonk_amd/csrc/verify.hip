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

// compressed short-Weierstrass point of ark-serialize 0.3 (proof.rs:98-155 wire format) -> affine (Montgomery);
// both base fields have p = 3 mod 4, so sqrt(a) = a^((p + 1) / 4)
template <class C>
static bool decompress(const uint8_t* b, Affine<typename C::Fq>* out, bool* is_inf) {
    using Q = typename C::Fq;
    const size_t nb = Q::N * 4;
    const uint8_t flags = b[nb - 1];
    if (flags & 0x40) {
        *is_inf = true;
        out->x = fe_zero<Q>();
        out->y = fe_zero<Q>();
        return true;
    }
    *is_inf = false;
    Fe<Q> x;
    if (!canonical_from_bytes<Q>(b, 2, &x)) return false;
    const Fe<Q> rhs = fe_add<Q>(fe_mul<Q>(fe_sqr<Q>(x), x), fe_from_u32<Q>(C::B));
    uint32_t e[Q::N];
    uint64_t carry = 1;   // (p + 1) / 4
    for (int i = 0; i < Q::N; ++i) {
        carry += Q::mod(i);
        e[i] = (uint32_t)carry;
        carry >>= 32;
    }
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < Q::N; ++i) e[i] = (e[i] >> 1) | (i + 1 < Q::N ? (e[i + 1] << 31) : ((uint32_t)carry << 31));
    Fe<Q> y = pow_words<Q>(rhs, e, Q::N);
    if (!fe_eq<Q>(fe_sqr<Q>(y), rhs)) return false;   // x is not on the curve
    if (is_larger_root<Q>(y) != ((flags & 0x80) != 0)) y = fe_neg<Q>(y);
    out->x = x;
    out->y = y;
    return true;
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
        HX r_commit = hostec::hx_identity<Q>();
        for (int k = 0; k < 13; ++k) r_commit = hostec::hx_add<Q>(r_commit, mul(to_hx(*pt[k]), sc[k]));
        static const char* EL[12] = {"a_eval", "b_eval", "c_eval", "sigma1_eval", "sigma2_eval", "z1_next_eval",
                                     "q_lookup_eval", "t_eval", "t_next_eval", "z2_next_eval", "h1_next_eval", "h2_eval"};
        for (int k = 0; k < 12; ++k) {
            uint8_t b[32];
            to_le(ev[k], b);
            tr.append_scalars(EL[k], b, 1, 32, true);
        }
        const F eta = challenge(tr, "eta");
        // ---- the two openings (proof.rs:420-500) ----
        Affine<Q> g;
        memcpy(g.x.v, in.g, nb);
        memcpy(g.y.v, in.g + L64, nb);
        const F w = root_of_unity<R>(log_n);
        auto pair = [&](const HX* commits, const F* values, int k, const F& z, const Affine<Q>& wit, int slot) {
            HX comb = hostec::hx_identity<Q>();
            F comb_v = fe_zero<R>(), ch = one;
            for (int i = 0; i < k; ++i) {
                comb = hostec::hx_add<Q>(comb, mul(commits[i], ch));
                comb_v = fe_add<R>(comb_v, fe_mul<R>(ch, values[i]));
                ch = fe_mul<R>(ch, eta);
            }
            HX L = hostec::hx_add<Q>(comb, mul(to_hx(g), fe_neg<R>(comb_v)));
            L = hostec::hx_add<Q>(L, mul(to_hx(wit), z));
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
        {
            const HX cs[9] = {r_commit, to_hx(cm[A]), to_hx(cm[B]), to_hx(cm[Cc]), to_hx(vk[S1]), to_hx(vk[S2]), to_hx(vk[QLOOKUP]),
                              to_hx(cm[T]), to_hx(cm[H2])};
            const F vs[9] = {r0, e_a, e_b, e_c, e_s1, e_s2, e_ql, e_t, e_h2};
            pair(cs, vs, 9, xi, cm[AW], 0);
        }
        {
            const HX cs[4] = {to_hx(cm[Z1]), to_hx(cm[Z2]), to_hx(cm[T]), to_hx(cm[H1])};
            const F vs[4] = {e_z1n, e_z2n, e_tn, e_h1n};
            pair(cs, vs, 4, fe_mul<R>(xi, w), cm[SAW], 1);
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
    }
    return ZKT_OK;
}

template <class C>
static int pairing_check_t(const uint64_t* g1, const uint64_t* g2, size_t n, int* is_one) {
    std::vector<typename pairing::Tower<C>::G1> ps;
    std::vector<typename pairing::Tower<C>::G2> qs;
    int rc = pairing_inputs<C>(g1, g2, n, ps, qs);
    if (rc) return rc;
    *is_one = pairing::Tower<C>::product_is_one(ps.data(), qs.data(), n) ? 1 : 0;
    return ZKT_OK;
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

// The whole of Proof::verify (proof.rs:285-503): zkt_verify_prepare, then for each opening e(L, h) * e(-W, beta h) == 1
template <class C>
static int verify_t(int curve_id, const zkt_verify_inputs* in, zkt_transcript* tr, const uint64_t* h, const uint64_t* beta_h,
                    int* accepted) {
    using Q = typename C::Fq;
    constexpr int L64 = Q::N / 2;
    uint64_t pairs[4 * 12];
    int inf[4];
    int rc = zkt_verify_prepare(curve_id, in, tr, pairs, inf);
    if (rc) return rc;
    *accepted = 1;
    for (int k = 0; k < 2 && *accepted; ++k) {
        uint64_t g1[2 * 12], g2[2 * 4 * 6];
        memcpy(g1, pairs + (size_t)(2 * k) * 2 * L64, 2 * L64 * 8);            // L
        Fe<Q> wy;                                                               // -W
        memcpy(g1 + 2 * L64, pairs + (size_t)(2 * k + 1) * 2 * L64, L64 * 8);
        memcpy(wy.v, pairs + (size_t)(2 * k + 1) * 2 * L64 + L64, L64 * 8);
        if (!inf[2 * k + 1]) wy = fe_neg<Q>(wy);
        memcpy(g1 + 3 * L64, wy.v, L64 * 8);
        memcpy(g2, h, 4 * L64 * 8);
        memcpy(g2 + 4 * L64, beta_h, 4 * L64 * 8);
        int one = 0;
        if ((rc = pairing_check_t<C>(g1, g2, 2, &one))) return rc;
        if (!one) *accepted = 0;    // Error::ProofVerificationError { step: k + 1 }
    }
    return ZKT_OK;
}

extern "C" int zkt_verify(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, const uint64_t* h_g2_mont,
                          const uint64_t* beta_h_g2_mont, int* accepted) {
    if (!in || !transcript || !h_g2_mont || !beta_h_g2_mont || !accepted) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return verify_t<Bn254Curve>(curve_id, in, transcript, h_g2_mont, beta_h_g2_mont, accepted);
    if (curve_id == ZKT_CURVE_BLS12_381) return verify_t<Bls381Curve>(curve_id, in, transcript, h_g2_mont, beta_h_g2_mont, accepted);
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
