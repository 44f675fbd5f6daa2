// Short-Weierstrass G1 (a = 0) arithmetic in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2, identity: ZZ = 0) over Fe<Fq>.
//
// Replaces the group operations ark-ec 0.3's VariableBaseMSM performs through
// GroupProjective::{add_assign_mixed, add_assign, double_in_place} (reached from
// plonk-core/src/commitment.rs:42,78 and kzg10 commit/open).  The formulas differ (XYZZ needs
// 10 multiplications per mixed addition instead of 11 and no inversion until the very end); the
// resulting group element, hence the affine output, is identical.
#pragma once
#include "fp.hpp"
#include "fx.hpp"
#include "hostinv.hpp"

namespace zkt {

struct Bn254Curve {
    using Fq = Bn254Fq;
    using Fr = Bn254Fr;
    static constexpr int ID = 0;
    static constexpr uint32_t B = 3;   // y^2 = x^3 + 3
};
struct Bls381Curve {
    using Fq = Bls381Fq;
    using Fr = Bls381Fr;
    static constexpr int ID = 1;
    static constexpr uint32_t B = 4;   // y^2 = x^3 + 4
};

template <class Q>
struct Affine {  // (0, 0) encodes the point at infinity on the C-ABI and in the tables
    Fe<Q> x, y;
};
template <class Q>
struct Xyzz {
    Fe<Q> x, y, zz, zzz;
};

template <class Q>
ZKT_HD bool aff_is_inf(const Affine<Q>& p) {
    return fe_is_zero<Q>(p.x) && fe_is_zero<Q>(p.y);
}
template <class Q>
ZKT_HD Xyzz<Q> xyzz_identity() {
    Xyzz<Q> r;
    r.x = fe_zero<Q>();
    r.y = fe_zero<Q>();
    r.zz = fe_zero<Q>();
    r.zzz = fe_zero<Q>();
    return r;
}
template <class Q>
ZKT_HD bool xyzz_is_identity(const Xyzz<Q>& p) {
    return fe_is_zero<Q>(p.zz);
}
template <class Q>
ZKT_HD Xyzz<Q> xyzz_from_affine(const Affine<Q>& p) {
    Xyzz<Q> r;
    if (aff_is_inf<Q>(p)) return xyzz_identity<Q>();
    r.x = p.x;
    r.y = p.y;
    r.zz = fe_one<Q>();
    r.zzz = fe_one<Q>();
    return r;
}

// dbl-2008-s-1
template <class Q>
ZKT_HD Xyzz<Q> xyzz_double(const Xyzz<Q>& p) {
    if (xyzz_is_identity<Q>(p)) return p;
    Fe<Q> u = fe_dbl<Q>(p.y);
    Fe<Q> v = fe_sqr<Q>(u);
    Fe<Q> w = fe_mul<Q>(u, v);
    Fe<Q> s = fe_mul<Q>(p.x, v);
    Fe<Q> xx = fe_sqr<Q>(p.x);
    Fe<Q> m = fe_add<Q>(fe_dbl<Q>(xx), xx);
    Xyzz<Q> r;
    r.x = fe_sub<Q>(fe_sqr<Q>(m), fe_dbl<Q>(s));
    r.y = fe_sub<Q>(fe_mul<Q>(m, fe_sub<Q>(s, r.x)), fe_mul<Q>(w, p.y));
    r.zz = fe_mul<Q>(v, p.zz);
    r.zzz = fe_mul<Q>(w, p.zzz);
    return r;
}

// mdbl-2008-s: doubling of an affine point
template <class Q>
ZKT_HD Xyzz<Q> xyzz_double_affine(const Affine<Q>& p) {
    Fe<Q> u = fe_dbl<Q>(p.y);
    Fe<Q> v = fe_sqr<Q>(u);
    Fe<Q> w = fe_mul<Q>(u, v);
    Fe<Q> s = fe_mul<Q>(p.x, v);
    Fe<Q> xx = fe_sqr<Q>(p.x);
    Fe<Q> m = fe_add<Q>(fe_dbl<Q>(xx), xx);
    Xyzz<Q> r;
    r.x = fe_sub<Q>(fe_sqr<Q>(m), fe_dbl<Q>(s));
    r.y = fe_sub<Q>(fe_mul<Q>(m, fe_sub<Q>(s, r.x)), fe_mul<Q>(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// madd-2008-s with the exceptional cases (identity accumulator, P == Q, P == -Q) handled.
// `q` must not be the point at infinity (callers filter (0,0)).
template <class Q>
ZKT_HD Xyzz<Q> xyzz_add_mixed(const Xyzz<Q>& p, const Affine<Q>& q) {
    if (xyzz_is_identity<Q>(p)) {
        Xyzz<Q> r;
        r.x = q.x;
        r.y = q.y;
        r.zz = fe_one<Q>();
        r.zzz = fe_one<Q>();
        return r;
    }
    Fe<Q> u2 = fe_mul<Q>(q.x, p.zz);
    Fe<Q> s2 = fe_mul<Q>(q.y, p.zzz);
    Fe<Q> pp_ = fe_sub<Q>(u2, p.x);
    Fe<Q> rr = fe_sub<Q>(s2, p.y);
    if (fe_is_zero<Q>(pp_)) {
        if (fe_is_zero<Q>(rr)) return xyzz_double_affine<Q>(q);
        return xyzz_identity<Q>();
    }
    Fe<Q> pp = fe_sqr<Q>(pp_);
    Fe<Q> ppp = fe_mul<Q>(pp_, pp);
    Fe<Q> qq = fe_mul<Q>(p.x, pp);
    Xyzz<Q> r;
    r.x = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(rr), ppp), fe_dbl<Q>(qq));
    r.y = fe_sub<Q>(fe_mul<Q>(rr, fe_sub<Q>(qq, r.x)), fe_mul<Q>(p.y, ppp));
    r.zz = fe_mul<Q>(p.zz, pp);
    r.zzz = fe_mul<Q>(p.zzz, ppp);
    return r;
}

// add-2008-s
template <class Q>
ZKT_HD Xyzz<Q> xyzz_add(const Xyzz<Q>& p, const Xyzz<Q>& q) {
    if (xyzz_is_identity<Q>(p)) return q;
    if (xyzz_is_identity<Q>(q)) return p;
    Fe<Q> u1 = fe_mul<Q>(p.x, q.zz);
    Fe<Q> u2 = fe_mul<Q>(q.x, p.zz);
    Fe<Q> s1 = fe_mul<Q>(p.y, q.zzz);
    Fe<Q> s2 = fe_mul<Q>(q.y, p.zzz);
    Fe<Q> pp_ = fe_sub<Q>(u2, u1);
    Fe<Q> rr = fe_sub<Q>(s2, s1);
    if (fe_is_zero<Q>(pp_)) {
        if (fe_is_zero<Q>(rr)) return xyzz_double<Q>(p);
        return xyzz_identity<Q>();
    }
    Fe<Q> pp = fe_sqr<Q>(pp_);
    Fe<Q> ppp = fe_mul<Q>(pp_, pp);
    Fe<Q> qq = fe_mul<Q>(u1, pp);
    Xyzz<Q> r;
    r.x = fe_sub<Q>(fe_sub<Q>(fe_sqr<Q>(rr), ppp), fe_dbl<Q>(qq));
    r.y = fe_sub<Q>(fe_mul<Q>(rr, fe_sub<Q>(qq, r.x)), fe_mul<Q>(s1, ppp));
    r.zz = fe_mul<Q>(fe_mul<Q>(p.zz, q.zz), pp);
    r.zzz = fe_mul<Q>(fe_mul<Q>(p.zzz, q.zzz), ppp);
    return r;
}

template <class Q>
ZKT_HD Affine<Q> xyzz_to_affine(const Xyzz<Q>& p) {
    Affine<Q> r;
    if (xyzz_is_identity<Q>(p)) {
        r.x = fe_zero<Q>();
        r.y = fe_zero<Q>();
        return r;
    }
    Fe<Q> inv = fe_inv<Q>(fe_mul<Q>(p.zz, p.zzz));
    r.x = fe_mul<Q>(p.x, fe_mul<Q>(inv, p.zzz));  // X / ZZ
    r.y = fe_mul<Q>(p.y, fe_mul<Q>(inv, p.zz));   // Y / ZZZ
    return r;
}

// host-side normalisation (the prover needs the affine commitment for the transcript): same arithmetic with the
// binary-GCD inversion of hostinv.hpp instead of the Fermat ladder the kernels use
template <class Q>
inline Affine<Q> xyzz_to_affine_host(const Xyzz<Q>& p) {
    Affine<Q> r;
    if (xyzz_is_identity<Q>(p)) {
        r.x = fe_zero<Q>();
        r.y = fe_zero<Q>();
        return r;
    }
    Fe<Q> inv = fe_inv_host<Q>(fe_mul<Q>(p.zz, p.zzz));
    r.x = fe_mul<Q>(p.x, fe_mul<Q>(inv, p.zzz));  // X / ZZ
    r.y = fe_mul<Q>(p.y, fe_mul<Q>(inv, p.zz));   // Y / ZZZ
    return r;
}

template <class Q>
ZKT_D Affine<Q> aff_load(const Affine<Q>* p) {
    Affine<Q> r;
    r.x = fe_load<Q>(&p->x);
    r.y = fe_load<Q>(&p->y);
    return r;
}
template <class Q>
ZKT_D void aff_store(Affine<Q>* p, const Affine<Q>& a) {
    fe_store<Q>(&p->x, a.x);
    fe_store<Q>(&p->y, a.y);
}
template <class Q>
ZKT_D Xyzz<Q> xyzz_load(const Xyzz<Q>* p) {
    Xyzz<Q> r;
    r.x = fe_load<Q>(&p->x);
    r.y = fe_load<Q>(&p->y);
    r.zz = fe_load<Q>(&p->zz);
    r.zzz = fe_load<Q>(&p->zzz);
    return r;
}
template <class Q>
ZKT_D void xyzz_store(Xyzz<Q>* p, const Xyzz<Q>& a) {
    fe_store<Q>(&p->x, a.x);
    fe_store<Q>(&p->y, a.y);
    fe_store<Q>(&p->zz, a.zz);
    fe_store<Q>(&p->zzz, a.zzz);
}

}  // namespace zkt
