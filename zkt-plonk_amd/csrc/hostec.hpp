// Host-side G1 arithmetic on 64-bit limbs (XYZZ coordinates, arkworks' R = 2^(64 N) Montgomery form).
//
// The last stage of an MSM's bucket reduction is a chain of ~35 DEPENDENT curve operations (row sums, 16 doublings,
// a short tree): one wavefront runs such a chain at ~17 us per operation (latency bound, nothing to overlap), the
// host at ~0.35 us.  So the kernels stop at the masked partial sums (msm.hip k_msm_masked_sums), ~25-37 KiB cross
// PCIe, and the host -- which needs the affine commitment for the Fiat-Shamir transcript anyway -- finishes here.
#pragma once
#include "ec.hpp"

#include <cstring>

namespace zkt {
namespace hostec {

typedef unsigned __int128 u128;

template <class Q>
struct HF {
    static constexpr int N = Q::N / 2;
    uint64_t v[N];
};

template <class Q>
struct HParams {
    static constexpr int N = Q::N / 2;
    static constexpr uint64_t mod(int i) { return (uint64_t)Q::mod(2 * i) | ((uint64_t)Q::mod(2 * i + 1) << 32); }
    static constexpr uint64_t inv() {   // -p^-1 mod 2^64 (Newton on the odd low limb)
        uint64_t p0 = mod(0), x = 1;
        for (int i = 0; i < 6; ++i) x *= 2 - p0 * x;
        return (uint64_t)0 - x;
    }
};

template <class Q>
inline HF<Q> hf_from(const Fe<Q>& a) {
    HF<Q> r;
    memcpy(r.v, a.v, sizeof(r.v));
    return r;
}
template <class Q>
inline Fe<Q> hf_to(const HF<Q>& a) {
    Fe<Q> r;
    memcpy(r.v, a.v, sizeof(a.v));
    return r;
}
template <class Q>
inline bool hf_is_zero(const HF<Q>& a) {
    uint64_t o = 0;
    for (int i = 0; i < HF<Q>::N; ++i) o |= a.v[i];
    return o == 0;
}
template <class Q>
inline bool hf_geq_p(const uint64_t* a) {
    for (int i = HF<Q>::N - 1; i >= 0; --i) {
        if (a[i] != HParams<Q>::mod(i)) return a[i] > HParams<Q>::mod(i);
    }
    return true;
}
template <class Q>
inline void hf_sub_p(uint64_t* a) {
    uint64_t borrow = 0;
    for (int i = 0; i < HF<Q>::N; ++i) {
        const u128 d = (u128)a[i] - HParams<Q>::mod(i) - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
template <class Q>
inline HF<Q> hf_add(const HF<Q>& a, const HF<Q>& b) {
    HF<Q> r;
    uint64_t carry = 0;
    for (int i = 0; i < HF<Q>::N; ++i) {
        const u128 s = (u128)a.v[i] + b.v[i] + carry;
        r.v[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    if (carry || hf_geq_p<Q>(r.v)) hf_sub_p<Q>(r.v);
    return r;
}
template <class Q>
inline HF<Q> hf_sub(const HF<Q>& a, const HF<Q>& b) {
    HF<Q> r;
    uint64_t borrow = 0;
    for (int i = 0; i < HF<Q>::N; ++i) {
        const u128 d = (u128)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < HF<Q>::N; ++i) {
            const u128 s = (u128)r.v[i] + HParams<Q>::mod(i) + carry;
            r.v[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    return r;
}
// CIOS Montgomery product, R = 2^(64 N)
template <class Q>
inline HF<Q> hf_mul(const HF<Q>& a, const HF<Q>& b) {
    constexpr int N = HF<Q>::N;
    constexpr uint64_t INV = HParams<Q>::inv();
    uint64_t t[N + 2] = {0};
    for (int i = 0; i < N; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < N; ++j) {
            const u128 x = (u128)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        u128 x = (u128)t[N] + c;
        t[N] = (uint64_t)x;
        t[N + 1] = (uint64_t)(x >> 64);
        const uint64_t m = t[0] * INV;
        x = (u128)m * HParams<Q>::mod(0) + t[0];
        c = (uint64_t)(x >> 64);
        for (int j = 1; j < N; ++j) {
            x = (u128)m * HParams<Q>::mod(j) + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        x = (u128)t[N] + c;
        t[N - 1] = (uint64_t)x;
        t[N] = t[N + 1] + (uint64_t)(x >> 64);
    }
    HF<Q> r;
    memcpy(r.v, t, sizeof(r.v));
    if (t[N] || hf_geq_p<Q>(r.v)) hf_sub_p<Q>(r.v);
    return r;
}

template <class Q>
struct HX {   // XYZZ, identity <=> zz == 0
    HF<Q> x, y, zz, zzz;
};
template <class Q>
inline HX<Q> hx_identity() {
    HX<Q> r;
    memset(&r, 0, sizeof(r));
    return r;
}
template <class Q>
inline HX<Q> hx_from(const Xyzz<Q>& p) {
    HX<Q> r;
    r.x = hf_from<Q>(p.x);
    r.y = hf_from<Q>(p.y);
    r.zz = hf_from<Q>(p.zz);
    r.zzz = hf_from<Q>(p.zzz);
    return r;
}
template <class Q>
inline Xyzz<Q> hx_to(const HX<Q>& p) {
    Xyzz<Q> r;
    r.x = hf_to<Q>(p.x);
    r.y = hf_to<Q>(p.y);
    r.zz = hf_to<Q>(p.zz);
    r.zzz = hf_to<Q>(p.zzz);
    return r;
}
// dbl-2008-s-1 (a = 0)
template <class Q>
inline HX<Q> hx_double(const HX<Q>& p) {
    if (hf_is_zero<Q>(p.zz)) return p;
    const HF<Q> u = hf_add<Q>(p.y, p.y);
    if (hf_is_zero<Q>(u)) return hx_identity<Q>();
    const HF<Q> v = hf_mul<Q>(u, u), w = hf_mul<Q>(u, v), s = hf_mul<Q>(p.x, v);
    const HF<Q> xx = hf_mul<Q>(p.x, p.x);
    const HF<Q> m = hf_add<Q>(hf_add<Q>(xx, xx), xx);
    HX<Q> r;
    r.x = hf_sub<Q>(hf_sub<Q>(hf_mul<Q>(m, m), s), s);
    r.y = hf_sub<Q>(hf_mul<Q>(m, hf_sub<Q>(s, r.x)), hf_mul<Q>(w, p.y));
    r.zz = hf_mul<Q>(v, p.zz);
    r.zzz = hf_mul<Q>(w, p.zzz);
    return r;
}
// add-2008-s
template <class Q>
inline HX<Q> hx_add(const HX<Q>& p, const HX<Q>& q) {
    if (hf_is_zero<Q>(p.zz)) return q;
    if (hf_is_zero<Q>(q.zz)) return p;
    const HF<Q> u1 = hf_mul<Q>(p.x, q.zz), u2 = hf_mul<Q>(q.x, p.zz);
    const HF<Q> s1 = hf_mul<Q>(p.y, q.zzz), s2 = hf_mul<Q>(q.y, p.zzz);
    const HF<Q> pp_ = hf_sub<Q>(u2, u1), rr = hf_sub<Q>(s2, s1);
    if (hf_is_zero<Q>(pp_)) return hf_is_zero<Q>(rr) ? hx_double<Q>(p) : hx_identity<Q>();
    const HF<Q> pp = hf_mul<Q>(pp_, pp_), ppp = hf_mul<Q>(pp_, pp), qq = hf_mul<Q>(u1, pp);
    HX<Q> r;
    r.x = hf_sub<Q>(hf_sub<Q>(hf_sub<Q>(hf_mul<Q>(rr, rr), ppp), qq), qq);
    r.y = hf_sub<Q>(hf_mul<Q>(rr, hf_sub<Q>(qq, r.x)), hf_mul<Q>(s1, ppp));
    r.zz = hf_mul<Q>(hf_mul<Q>(p.zz, q.zz), pp);
    r.zzz = hf_mul<Q>(hf_mul<Q>(p.zzz, q.zzz), ppp);
    return r;
}

// sum_y 2^(e_y) * V_y with V_y = sum_j rows[y * nblk + j]; exps ascending.  Horner from the top exponent down.
template <class Q>
inline Xyzz<Q> weighted_row_sum(const Xyzz<Q>* rows, int nrows, int nblk, const int* exps) {
    HX<Q> acc = hx_identity<Q>();
    int e = nrows ? exps[nrows - 1] : 0;
    for (int y = nrows - 1; y >= 0; --y) {
        for (; e > exps[y]; --e) acc = hx_double<Q>(acc);
        HX<Q> v = hx_identity<Q>();
        for (int j = 0; j < nblk; ++j) v = hx_add<Q>(v, hx_from<Q>(rows[(size_t)y * nblk + j]));
        acc = hx_add<Q>(acc, v);
    }
    for (; e > 0; --e) acc = hx_double<Q>(acc);
    return hx_to<Q>(acc);
}

}  // namespace hostec
}  // namespace zkt
