// Host-side reduced Tate pairing on BN254 and BLS12-381 (the last step of verification).
//
// The reference leaves the check e(L, h) == e(W, beta h) to ark-ec's optimal ate pairing (SonicKZG10::check, reached from
// plonk-core/src/proof_system/proof.rs:420-500).  A verifier only asks whether a PRODUCT of pairings is one, and every
// non-degenerate bilinear pairing on G1 x G2 answers that identically, so the simplest one is computed here:
//     t(P, Q) = f_{r,P}(psi(Q)) ^ ((p^12 - 1) / r),   P in G1 = E(Fq)[r],  Q in G2 = E'(Fq2)[r]
// Tower Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u (BN254) / 1 + u (BLS12-381);
// twists: BN254 D-type psi(x, y) = (x w^2, y w^3), BLS12-381 M-type psi(x, y) = (x / w^2, y / w^3).  Miller loop over the
// bits of r with affine lines through the multiples of P (Fq arithmetic on 64-bit limbs, hostec.hpp) evaluated at psi(Q);
// vertical lines die in the final exponentiation, which is conj(f)/f, then a power by p^2 + 1, then one by
// (p^4 - p^2 + 1)/r (embedded constant).  Two pairings cost ~10 ms of host time: a device has nothing to add.
#pragma once
#include "hostec.hpp"
#include "hostinv.hpp"

#include <vector>

namespace zkt {
namespace pairing {

using hostec::HF;

template <class Q>
struct Ops {
    typedef HF<Q> F;
    static F zero() { F r; memset(r.v, 0, sizeof(r.v)); return r; }
    static F one() { return hostec::hf_from<Q>(fe_one<Q>()); }
    static F from_u32(uint32_t x) { return hostec::hf_from<Q>(fe_from_u32<Q>(x)); }
    static F add(const F& a, const F& b) { return hostec::hf_add<Q>(a, b); }
    static F sub(const F& a, const F& b) { return hostec::hf_sub<Q>(a, b); }
    static F neg(const F& a) { return hostec::hf_sub<Q>(zero(), a); }
    static F mul(const F& a, const F& b) { return hostec::hf_mul<Q>(a, b); }
    static F inv(const F& a) { return hostec::hf_from<Q>(fe_inv_host<Q>(hostec::hf_to<Q>(a))); }
    static bool is_zero(const F& a) { return hostec::hf_is_zero<Q>(a); }
    static bool eq(const F& a, const F& b) { return memcmp(a.v, b.v, sizeof(a.v)) == 0; }
};

template <class Q> struct F2 { HF<Q> c0, c1; };
template <class Q> struct F6 { F2<Q> c0, c1, c2; };
template <class Q> struct F12 { F6<Q> c0, c1; };

template <class C>
struct Tower {
    typedef typename C::Fq Q;
    typedef Ops<Q> O;
    typedef HF<Q> F;
    typedef F2<Q> E2;
    typedef F6<Q> E6;
    typedef F12<Q> E12;

    // xi = 9 + u (BN254), 1 + u (BLS12-381)
    static E2 xi() { return E2{O::from_u32(C::ID == 0 ? 9 : 1), O::one()}; }

    static E2 z2() { return E2{O::zero(), O::zero()}; }
    static E2 o2() { return E2{O::one(), O::zero()}; }
    static E2 add2(const E2& a, const E2& b) { return E2{O::add(a.c0, b.c0), O::add(a.c1, b.c1)}; }
    static E2 sub2(const E2& a, const E2& b) { return E2{O::sub(a.c0, b.c0), O::sub(a.c1, b.c1)}; }
    static E2 neg2(const E2& a) { return E2{O::neg(a.c0), O::neg(a.c1)}; }
    static E2 mul2(const E2& a, const E2& b) {   // (a0 + a1 u)(b0 + b1 u), u^2 = -1 (Karatsuba)
        const F t0 = O::mul(a.c0, b.c0), t1 = O::mul(a.c1, b.c1);
        const F s = O::mul(O::add(a.c0, a.c1), O::add(b.c0, b.c1));
        return E2{O::sub(t0, t1), O::sub(O::sub(s, t0), t1)};
    }
    static E2 scal2(const E2& a, const F& s) { return E2{O::mul(a.c0, s), O::mul(a.c1, s)}; }
    static E2 inv2(const E2& a) {
        const F d = O::inv(O::add(O::mul(a.c0, a.c0), O::mul(a.c1, a.c1)));
        return E2{O::mul(a.c0, d), O::neg(O::mul(a.c1, d))};
    }
    static bool is_zero2(const E2& a) { return O::is_zero(a.c0) && O::is_zero(a.c1); }

    static E6 z6() { return E6{z2(), z2(), z2()}; }
    static E6 o6() { return E6{o2(), z2(), z2()}; }
    static E6 add6(const E6& a, const E6& b) { return E6{add2(a.c0, b.c0), add2(a.c1, b.c1), add2(a.c2, b.c2)}; }
    static E6 sub6(const E6& a, const E6& b) { return E6{sub2(a.c0, b.c0), sub2(a.c1, b.c1), sub2(a.c2, b.c2)}; }
    static E6 neg6(const E6& a) { return E6{neg2(a.c0), neg2(a.c1), neg2(a.c2)}; }
    static E6 mul6(const E6& a, const E6& b) {   // schoolbook in v, v^3 = xi
        const E2 x = xi();
        const E2 t0 = mul2(a.c0, b.c0);
        const E2 t1 = add2(mul2(a.c0, b.c1), mul2(a.c1, b.c0));
        const E2 t2 = add2(add2(mul2(a.c0, b.c2), mul2(a.c1, b.c1)), mul2(a.c2, b.c0));
        const E2 t3 = add2(mul2(a.c1, b.c2), mul2(a.c2, b.c1));
        const E2 t4 = mul2(a.c2, b.c2);
        return E6{add2(t0, mul2(x, t3)), add2(t1, mul2(x, t4)), t2};
    }
    static E6 mulv6(const E6& a) { return E6{mul2(xi(), a.c2), a.c0, a.c1}; }
    static E6 scal6(const E6& a, const F& s) { return E6{scal2(a.c0, s), scal2(a.c1, s), scal2(a.c2, s)}; }
    static E6 inv6(const E6& a) {
        const E2 x = xi();
        const E2 c0 = sub2(mul2(a.c0, a.c0), mul2(x, mul2(a.c1, a.c2)));
        const E2 c1 = sub2(mul2(x, mul2(a.c2, a.c2)), mul2(a.c0, a.c1));
        const E2 c2 = sub2(mul2(a.c1, a.c1), mul2(a.c0, a.c2));
        const E2 t = add2(mul2(a.c0, c0), mul2(x, add2(mul2(a.c2, c1), mul2(a.c1, c2))));
        const E2 ti = inv2(t);
        return E6{mul2(c0, ti), mul2(c1, ti), mul2(c2, ti)};
    }

    static E12 o12() { return E12{o6(), z6()}; }
    static E12 mul12(const E12& a, const E12& b) {
        const E6 a0b0 = mul6(a.c0, b.c0), a1b1 = mul6(a.c1, b.c1);
        const E6 c1 = sub6(sub6(mul6(add6(a.c0, a.c1), add6(b.c0, b.c1)), a0b0), a1b1);
        return E12{add6(a0b0, mulv6(a1b1)), c1};
    }
    static E12 sub12(const E12& a, const E12& b) { return E12{sub6(a.c0, b.c0), sub6(a.c1, b.c1)}; }
    static E12 scal12(const E12& a, const F& s) { return E12{scal6(a.c0, s), scal6(a.c1, s)}; }
    static E12 conj12(const E12& a) { return E12{a.c0, neg6(a.c1)}; }
    static E12 inv12(const E12& a) {
        const E6 t = inv6(sub6(mul6(a.c0, a.c0), mulv6(mul6(a.c1, a.c1))));
        return E12{mul6(a.c0, t), neg6(mul6(a.c1, t))};
    }
    static E12 from_f(const F& s) { return E12{E6{E2{s, O::zero()}, z2(), z2()}, z6()}; }
    static E12 from_f2(const E2& x) { return E12{E6{x, z2(), z2()}, z6()}; }
    static bool is_one12(const E12& a) {
        const E12 o = o12();
        return memcmp(&a, &o, sizeof(a)) == 0;   // every limb is canonical (fully reduced)
    }
    // power by a little-endian array of 64-bit words
    static E12 pow12(const E12& a, const uint64_t* e, int nwords) {
        E12 r = o12();
        bool started = false;
        for (int i = nwords * 64 - 1; i >= 0; --i) {
            if (started) r = mul12(r, r);
            if ((e[i / 64] >> (i % 64)) & 1) {
                r = started ? mul12(r, a) : a;
                started = true;
            }
        }
        return r;
    }

    // (p^4 - p^2 + 1) / r, little-endian 64-bit words (computed with Python big integers from the published moduli)
    static const uint64_t* hard_exponent(int* nwords) {
        static const uint64_t bn[12] = {
            0xe81bb482ccdf42b1ULL, 0x5abf5cc4f49c36d4ULL, 0xf1154e7e1da014fdULL, 0xdcc7b44c87cdbacfULL,
            0xaaa441e3954bcf8aULL, 0x6b887d56d5095f23ULL, 0x79581e16f3fd90c6ULL, 0x3b1b1355d189227dULL,
            0x4e529a5861876f6bULL, 0x6c0eb522d5b12278ULL, 0x331ec15183177fafULL, 0x01baaa710b0759adULL};
        static const uint64_t bls[20] = {
            0xe516c3f438e3ba79ULL, 0xfa9912aae208ccf1ULL, 0x905ce937335d5b68ULL, 0xc71a2629b0dea236ULL,
            0x83774940996754c8ULL, 0x21d160aeb6a1e799ULL, 0x2ed0b283ed237db4ULL, 0x915c97f36c6f1821ULL,
            0x67f17fcbde783765ULL, 0x2378b9039096d1b7ULL, 0x7988f8761bdc51dcULL, 0x2076995003fc77a1ULL,
            0x827eca0ba621315bULL, 0xe5a72bce8d63cb9fULL, 0xf68f7764c28b6f8aULL, 0x2f230063cf081517ULL,
            0x94506632528d6a9aULL, 0xd3cde88eeb996ca3ULL, 0xc0bd38c3195c899eULL, 0x000f686b3d807d01ULL};
        if (C::ID == 0) {
            *nwords = 12;
            return bn;
        }
        *nwords = 20;
        return bls;
    }

    // f^((p^12 - 1) / r) = ((conj(f) / f)^(p^2 + 1))^((p^4 - p^2 + 1) / r)
    static E12 final_exponentiation(const E12& f) {
        const E12 f1 = mul12(conj12(f), inv12(f));
        constexpr int N = Q::N / 2;
        uint64_t p[N], e[2 * N + 1] = {0};   // p^2 + 1
        for (int i = 0; i < N; ++i) p[i] = hostec::HParams<Q>::mod(i);
        for (int i = 0; i < N; ++i) {
            uint64_t carry = 0;
            for (int j = 0; j < N; ++j) {
                const hostec::u128 x = (hostec::u128)p[i] * p[j] + e[i + j] + carry;
                e[i + j] = (uint64_t)x;
                carry = (uint64_t)(x >> 64);
            }
            e[i + N] += carry;
        }
        for (int i = 0; i < 2 * N + 1; ++i)
            if (++e[i] != 0) break;
        const E12 f2 = pow12(f1, e, 2 * N + 1);
        int nw = 0;
        const uint64_t* h = hard_exponent(&nw);
        return pow12(f2, h, nw);
    }

    struct G1 { F x, y; bool inf; };
    struct G2 { E2 x, y; bool inf; };

    // f_{r,P}(psi(Q))
    static E12 miller(const G1& P, const G2& Qt) {
        if (P.inf || Qt.inf) return o12();
        // untwist: w^2 = v, w^3 = v w  (D-type: multiply, M-type: divide)
        E12 w2 = E12{E6{z2(), o2(), z2()}, z6()};
        E12 w3 = E12{z6(), E6{z2(), o2(), z2()}};
        if (C::ID != 0) {
            w2 = inv12(w2);
            w3 = inv12(w3);
        }
        const E12 xq = mul12(from_f2(Qt.x), w2), yq = mul12(from_f2(Qt.y), w3);
        E12 f = o12();
        F tx = P.x, ty = P.y;
        bool tinf = false;
        const F three = O::from_u32(3);
        auto line = [&](const F& lam) {   // (y_Q - y_T) - lam (x_Q - x_T)
            return sub12(sub12(yq, from_f(ty)), scal12(sub12(xq, from_f(tx)), lam));
        };
        typedef typename C::Fr R;
        int top = R::N * 32 - 1;
        while (!((R::mod(top / 32) >> (top % 32)) & 1u)) --top;
        for (int i = top - 1; i >= 0; --i) {
            // doubling step (T never has order two: y_T != 0)
            F lam = O::mul(O::mul(three, O::mul(tx, tx)), O::inv(O::add(ty, ty)));
            f = mul12(mul12(f, f), line(lam));
            F nx = O::sub(O::sub(O::mul(lam, lam), tx), tx);
            ty = O::sub(O::mul(lam, O::sub(tx, nx)), ty);
            tx = nx;
            if ((R::mod(i / 32) >> (i % 32)) & 1u) {
                if (O::eq(tx, P.x)) {   // T = -P (only at the very end): vertical chord, value in Fq6, dies in the final power
                    tinf = true;
                    continue;
                }
                lam = O::mul(O::sub(P.y, ty), O::inv(O::sub(P.x, tx)));
                f = mul12(f, line(lam));
                nx = O::sub(O::sub(O::mul(lam, lam), tx), P.x);
                ty = O::sub(O::mul(lam, O::sub(tx, nx)), ty);
                tx = nx;
            }
        }
        (void)tinf;
        return f;
    }

    static bool product_is_one(const G1* ps, const G2* qs, size_t n) {
        E12 f = o12();
        for (size_t i = 0; i < n; ++i) f = mul12(f, miller(ps[i], qs[i]));
        return is_one12(final_exponentiation(f));
    }

    // membership checks for untrusted inputs: on the curve / twist (the subgroup check of G2 is the caller's: h and
    // beta h come from the trusted VerifierKey)
    static bool g1_on_curve(const G1& P) {
        if (P.inf) return true;
        return O::eq(O::mul(P.y, P.y), O::add(O::mul(O::mul(P.x, P.x), P.x), O::from_u32(C::B)));
    }
    static bool g2_on_twist(const G2& Qt) {
        if (Qt.inf) return true;
        const E2 b{O::from_u32(C::B), O::zero()};
        const E2 bt = C::ID == 0 ? mul2(b, inv2(xi())) : mul2(b, xi());
        const E2 d = sub2(mul2(Qt.y, Qt.y), add2(mul2(mul2(Qt.x, Qt.x), Qt.x), bt));
        return is_zero2(d);
    }
};

}  // namespace pairing
}  // namespace zkt
