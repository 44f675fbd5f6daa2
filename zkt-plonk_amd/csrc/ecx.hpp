// G1 arithmetic on lazily reduced 29-bit-limb field elements (fx.hpp), XYZZ coordinates.
// Used by the MSM kernels; coordinates are in R' = 2^(29 L) Montgomery form throughout (the window
// table is converted once at load time, the single result once at the end).
//
// Value bounds carried by an XyzzX between operations (p = base-field modulus):
//     X < 8p,  Y < 4p,  ZZ < 2p,  ZZZ < 2p,   all limbs normalised.
// Every product below satisfies a*b < 128 p^2 <= R' * p (BN254 Fq: R'/p ~ 168; BLS12-381 Fq: 2^25).
#pragma once
#include "ec.hpp"
#include "fx.hpp"

namespace zkt {

template <class Q>
struct AffineX {      // canonical coordinates (< p), R' form
    Fx<Q> x, y;
};
template <class Q>
struct XyzzX {
    Fx<Q> x, y, zz, zzz;
    bool inf;         // identity flag kept beside the lazily reduced coordinates
};

template <class Q>
ZKT_HD XyzzX<Q> xx_identity() {
    XyzzX<Q> r;
    r.x = fx_zero<Q>();
    r.y = fx_zero<Q>();
    r.zz = fx_zero<Q>();
    r.zzz = fx_zero<Q>();
    r.inf = true;
    return r;
}

// zero test for a product output (value < 2p, normalised): 0 or p
template <class Q>
ZKT_HD bool fx_is_zero_lt2p(const Fx<Q>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < FxP<Q>::L; ++i) {
        z |= a.l[i];
        e |= a.l[i] ^ FxP<Q>::mod(i);
    }
    return z == 0 || e == 0;
}

// mdbl-2008-s on a canonical affine point
template <class Q>
ZKT_HD XyzzX<Q> xx_double_affine(const AffineX<Q>& p) {
    XyzzX<Q> r;
    const Fx<Q> u = fx_dbl<Q>(p.y);                    // < 2p
    const Fx<Q> v = fx_sqr<Q>(u);
    const Fx<Q> w = fx_mul<Q>(u, v);
    const Fx<Q> s = fx_mul<Q>(p.x, v);
    const Fx<Q> xx = fx_sqr<Q>(p.x);
    const Fx<Q> m = fx_add<Q>(fx_dbl<Q>(xx), xx);      // < 6p
    r.x = fx_sub<Q, 4>(fx_sqr<Q>(m), fx_dbl<Q>(s));    // < 6p
    r.y = fx_sub<Q, 2>(fx_mul<Q>(m, fx_sub<Q, 8>(s, r.x)), fx_mul<Q>(w, p.y));  // < 4p
    r.zz = v;
    r.zzz = w;
    r.inf = false;
    return r;
}

// dbl-2008-s-1
template <class Q>
ZKT_HD XyzzX<Q> xx_double(const XyzzX<Q>& p) {
    if (p.inf) return p;
    XyzzX<Q> r;
    const Fx<Q> u = fx_dbl<Q>(p.y);                    // < 8p
    const Fx<Q> v = fx_sqr<Q>(u);                      // 64 p^2
    const Fx<Q> w = fx_mul<Q>(u, v);
    const Fx<Q> s = fx_mul<Q>(p.x, v);
    const Fx<Q> xx = fx_sqr<Q>(p.x);                   // 64 p^2
    const Fx<Q> m = fx_add<Q>(fx_dbl<Q>(xx), xx);      // < 6p
    r.x = fx_sub<Q, 4>(fx_sqr<Q>(m), fx_dbl<Q>(s));    // < 6p
    r.y = fx_sub<Q, 2>(fx_mul<Q>(m, fx_sub<Q, 8>(s, r.x)), fx_mul<Q>(w, p.y));  // 6p * 10p ; < 4p
    r.zz = fx_mul<Q>(v, p.zz);
    r.zzz = fx_mul<Q>(w, p.zzz);
    r.inf = false;
    return r;
}

// madd-2008-s; q canonical, not the point at infinity.  This is the body of the MSM accumulation loop:
// there (INL) its products are inlined, the two squarings use the squaring kernel and Y3 = R (Q - X3) - Y1 PPP
// is one double product with a single reduction; everything else calls the shared product.
template <class Q, bool INL = false>
ZKT_HD XyzzX<Q> xx_add_mixed(const XyzzX<Q>& p, const AffineX<Q>& q) {
    auto mul = [](const Fx<Q>& a, const Fx<Q>& b) { return INL ? fx_mul_inl<Q>(a, b) : fx_mul<Q>(a, b); };
    auto sqr = [](const Fx<Q>& a) { return INL ? fx_sqr_inl<Q>(a) : fx_mul<Q>(a, a); };
    if (p.inf) {
        XyzzX<Q> r;
        r.x = q.x;
        r.y = q.y;
        r.zz = fx_one<Q>();
        r.zzz = fx_one<Q>();
        r.inf = false;
        return r;
    }
    const Fx<Q> u2 = mul(q.x, p.zz);
    const Fx<Q> s2 = mul(q.y, p.zzz);
    const Fx<Q> pp_ = fx_sub<Q, 8>(u2, p.x);           // < 10p
    const Fx<Q> rr = fx_sub<Q, 4>(s2, p.y);            // < 6p
    const Fx<Q> pp = sqr(pp_);                         // 100 p^2
    const Fx<Q> rr2 = sqr(rr);
    if (fx_is_zero_lt2p<Q>(pp)) {                      // same x: P == Q or P == -Q
        if (fx_is_zero_lt2p<Q>(rr2)) return xx_double_affine<Q>(q);
        return xx_identity<Q>();
    }
    const Fx<Q> ppp = mul(pp_, pp);
    const Fx<Q> qq = mul(p.x, pp);
    XyzzX<Q> r;
    if (INL) {
        r.x = fx_sub2<Q, 6>(rr2, ppp, qq);                                       // rr2 + 6p - ppp - 2 qq < 8p
        // Y3 = R (Q - X3) - Y1 PPP as one double product.  (Q + 9p - X3) < 11p, (5p - Y1) < 5p:
        // 6p * 11p + 5p * 2p < R' p ; result < 2p.  With nine limbs the two differences skip their carry pass
        // (limbs <= 3 * 2^29: the 27 column terms stay below 63 * 2^58); fourteen limbs would overflow the column.
        if constexpr (FxP<Q>::L <= 9)
            r.y = fx_mul2_inl<Q>(rr, fx_sub_lazy<Q, 9>(qq, r.x), ppp, fx_sub_lazy<Q, 5>(fx_zero<Q>(), p.y));
        else
            r.y = fx_mul2_inl<Q>(rr, fx_sub<Q, 8>(qq, r.x), ppp, fx_sub<Q, 4>(fx_zero<Q>(), p.y));
    } else {
        r.x = fx_sub<Q, 4>(fx_sub<Q, 2>(rr2, ppp), fx_dbl<Q>(qq));               // < 8p
        r.y = fx_sub<Q, 2>(mul(rr, fx_sub<Q, 8>(qq, r.x)), mul(p.y, ppp));       // 6p * 10p ; < 4p
    }
    r.zz = mul(p.zz, pp);
    r.zzz = mul(p.zzz, ppp);
    r.inf = false;
    return r;
}

// add-2008-s.  INL: every product inlined (the MSM's bucket reduction: a chain of ~25 dependent additions on one wave per
// SIMD, where an addition's latency is its instruction count and each call into the shared product costs ~60 of them)
template <class Q, bool INL = false>
ZKT_HD XyzzX<Q> xx_add(const XyzzX<Q>& p, const XyzzX<Q>& q) {
    auto mul = [](const Fx<Q>& a, const Fx<Q>& b) { return INL ? fx_mul_inl<Q>(a, b) : fx_mul<Q>(a, b); };
    auto sqr = [](const Fx<Q>& a) { return INL ? fx_sqr_inl<Q>(a) : fx_mul<Q>(a, a); };
    if (p.inf) return q;
    if (q.inf) return p;
    const Fx<Q> u1 = mul(p.x, q.zz);
    const Fx<Q> u2 = mul(q.x, p.zz);
    const Fx<Q> s1 = mul(p.y, q.zzz);
    const Fx<Q> s2 = mul(q.y, p.zzz);
    const Fx<Q> pp_ = fx_sub<Q, 2>(u2, u1);            // < 4p
    const Fx<Q> rr = fx_sub<Q, 2>(s2, s1);             // < 4p
    const Fx<Q> pp = sqr(pp_);
    if (fx_is_zero_lt2p<Q>(pp)) {
        if (fx_is_zero_lt2p<Q>(sqr(rr))) return xx_double<Q>(p);
        return xx_identity<Q>();
    }
    const Fx<Q> ppp = mul(pp_, pp);
    const Fx<Q> qq = mul(u1, pp);
    XyzzX<Q> r;
    r.x = fx_sub<Q, 4>(fx_sub<Q, 2>(sqr(rr), ppp), fx_dbl<Q>(qq));        // < 8p
    r.y = fx_sub<Q, 2>(mul(rr, fx_sub<Q, 8>(qq, r.x)), mul(s1, ppp));  // < 4p
    r.zz = mul(mul(p.zz, q.zz), pp);
    r.zzz = mul(mul(p.zzz, q.zzz), ppp);
    r.inf = false;
    return r;
}

// ---- memory forms: canonical packed words (Fe containers), R' Montgomery form ----------------------
template <class Q>
ZKT_D AffineX<Q> affx_load(const Affine<Q>* p, bool* is_inf) {
    const Fe<Q> x = fe_load<Q>(&p->x), y = fe_load<Q>(&p->y);
    *is_inf = fe_is_zero<Q>(x) && fe_is_zero<Q>(y);
    AffineX<Q> r;
    r.x = fx_unpack<Q>(x);
    r.y = fx_unpack<Q>(y);
    return r;
}
template <class Q>
ZKT_D XyzzX<Q> xx_load(const Xyzz<Q>* p) {
    XyzzX<Q> r;
    const Fe<Q> zz = fe_load<Q>(&p->zz);
    r.inf = fe_is_zero<Q>(zz);
    r.x = fx_unpack<Q>(fe_load<Q>(&p->x));
    r.y = fx_unpack<Q>(fe_load<Q>(&p->y));
    r.zz = fx_unpack<Q>(zz);
    r.zzz = fx_unpack<Q>(fe_load<Q>(&p->zzz));
    return r;
}
template <class Q>
ZKT_D void xx_store(Xyzz<Q>* p, const XyzzX<Q>& a) {
    if (a.inf) {
        const Fe<Q> z = fe_zero<Q>();
        fe_store<Q>(&p->x, z);
        fe_store<Q>(&p->y, z);
        fe_store<Q>(&p->zz, z);
        fe_store<Q>(&p->zzz, z);
        return;
    }
    fe_store<Q>(&p->x, fx_pack<Q>(fx_canon<Q>(a.x)));
    fe_store<Q>(&p->y, fx_pack<Q>(fx_canon<Q>(a.y)));
    fe_store<Q>(&p->zz, fx_pack<Q>(fx_canon<Q>(a.zz)));
    fe_store<Q>(&p->zzz, fx_pack<Q>(fx_canon<Q>(a.zzz)));
}
// Raw form: the 4 L limbs as they are (lazily reduced, bounds of XyzzX), all-zero = identity.  For intermediate
// sums that are read back once (the MSM's per-chunk pieces): no reduction, no packing on either side.
template <class Q>
struct alignas(16) XyzzRaw {
    uint32_t l[4 * FxP<Q>::L];
};
template <class Q>
ZKT_D void xx_store_raw(XyzzRaw<Q>* p, const XyzzX<Q>& a) {
    constexpr int L = FxP<Q>::L;
    static_assert((4 * L) % 4 == 0, "whole 16-byte words");
    uint32_t w[4 * L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        w[i] = a.inf ? 0u : a.x.l[i];
        w[L + i] = a.inf ? 0u : a.y.l[i];
        w[2 * L + i] = a.inf ? 0u : a.zz.l[i];
        w[3 * L + i] = a.inf ? 0u : a.zzz.l[i];
    }
    uint4* d = reinterpret_cast<uint4*>(p->l);
#pragma unroll
    for (int i = 0; i < L; ++i) d[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <class Q>
ZKT_D XyzzX<Q> xx_load_raw(const XyzzRaw<Q>* p) {
    constexpr int L = FxP<Q>::L;
    uint32_t w[4 * L];
    const uint4* s = reinterpret_cast<const uint4*>(p->l);
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint4 v = s[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    XyzzX<Q> r;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        r.x.l[i] = w[i];
        r.y.l[i] = w[L + i];
        r.zz.l[i] = w[2 * L + i];
        r.zzz.l[i] = w[3 * L + i];
        any |= w[2 * L + i];
    }
    r.inf = any == 0;
    return r;
}

// R' form -> arkworks R form (canonical packed), for the one point that leaves the MSM
template <class Q>
ZKT_D void xx_store_ark(Xyzz<Q>* p, const XyzzX<Q>& a) {
    if (a.inf) {
        const Fe<Q> z = fe_zero<Q>();
        fe_store<Q>(&p->x, z);
        fe_store<Q>(&p->y, z);
        fe_store<Q>(&p->zz, z);
        fe_store<Q>(&p->zzz, z);
        return;
    }
    fe_store<Q>(&p->x, fx_to_ark<Q>(a.x));
    fe_store<Q>(&p->y, fx_to_ark<Q>(a.y));
    fe_store<Q>(&p->zz, fx_to_ark<Q>(a.zz));
    fe_store<Q>(&p->zzz, fx_to_ark<Q>(a.zzz));
}

}  // namespace zkt
