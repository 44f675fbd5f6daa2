// Host-only modular inversion for the prover's control path (commitment normalisation, the grand-product
// denominators, 1/xi, L_1(xi)): binary extended Euclid on 32-bit words (HAC 14.61), ~2 * BITS cheap iterations
// instead of the ~1.5 * BITS Montgomery products of the Fermat ladder that the kernels use (fp.hpp fe_inv).
// A proof needs ~17 of these between GPU phases, so their latency is GPU idle time.
#pragma once
#include "fp.hpp"
#include "fx.hpp"

namespace zkt {
namespace hostinv {

template <int N>
struct Big {
    uint32_t w[N];
};
template <int N>
inline bool is_one(const Big<N>& a) {
    if (a.w[0] != 1u) return false;
    for (int i = 1; i < N; ++i)
        if (a.w[i]) return false;
    return true;
}
template <int N>
inline bool geq(const Big<N>& a, const Big<N>& b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    }
    return true;
}
template <int N>
inline uint32_t add(Big<N>& a, const Big<N>& b) {  // a += b, returns the carry out
    uint64_t c = 0;
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)a.w[i] + b.w[i];
        a.w[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}
template <int N>
inline void sub(Big<N>& a, const Big<N>& b) {  // a -= b (mod 2^(32N))
    uint64_t br = 0;
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)a.w[i] - b.w[i] - br;
        a.w[i] = (uint32_t)x;
        br = (x >> 63) & 1u;
    }
}
template <int N>
inline void shr1(Big<N>& a, uint32_t top) {  // (top : a) >> 1
    for (int i = 0; i < N - 1; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31);
    a.w[N - 1] = (a.w[N - 1] >> 1) | (top << 31);
}
// x / 2 mod p for x < p (p odd)
template <int N>
inline void half_mod(Big<N>& x, const Big<N>& p) {
    uint32_t top = 0;
    if (x.w[0] & 1u) top = add<N>(x, p);
    shr1<N>(x, top);
}
// x - y mod p for x, y < p
template <int N>
inline void sub_mod(Big<N>& x, const Big<N>& y, const Big<N>& p) {
    const bool wrap = !geq<N>(x, y);
    sub<N>(x, y);
    if (wrap) add<N>(x, p);
}

}  // namespace hostinv

// a^-1 in the same (arkworks R) Montgomery form; a != 0.  (a R)^-1 as a plain integer is a^-1 R^-1; two Montgomery
// products with R^2 bring it to a^-1 R.
template <class P>
inline Fe<P> fe_inv_host(const Fe<P>& a) {
    using namespace hostinv;
    constexpr int N = P::N;
    Big<N> u, v, x1, x2, p;
    if (fe_is_zero<P>(a)) return a;   // callers reject zero denominators themselves
    for (int i = 0; i < N; ++i) {
        u.w[i] = a.v[i];
        p.w[i] = P::mod(i);
        v.w[i] = p.w[i];
        x1.w[i] = 0;
        x2.w[i] = 0;
    }
    x1.w[0] = 1;
    while (!is_one<N>(u) && !is_one<N>(v)) {
        while ((u.w[0] & 1u) == 0) {
            shr1<N>(u, 0);
            half_mod<N>(x1, p);
        }
        while ((v.w[0] & 1u) == 0) {
            shr1<N>(v, 0);
            half_mod<N>(x2, p);
        }
        if (geq<N>(u, v)) {
            sub<N>(u, v);
            sub_mod<N>(x1, x2, p);
        } else {
            sub<N>(v, u);
            sub_mod<N>(x2, x1, p);
        }
    }
    const Big<N>& r = is_one<N>(u) ? x1 : x2;
    Fe<P> t, r2;
    for (int i = 0; i < N; ++i) {
        t.v[i] = r.w[i];
        r2.v[i] = P::r2(i);
    }
    return fe_mul<P>(fe_mul<P>(t, r2), r2);
}

}  // namespace zkt
