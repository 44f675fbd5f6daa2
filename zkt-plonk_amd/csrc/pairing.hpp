// Host-side pairing check on BN254 and BLS12-381: the last step of verification.
//
// The reference leaves e(L, h) == e(W, beta h) to ark-ec's optimal ate pairing (SonicKZG10::check, reached from
// plonk-core/src/proof_system/proof.rs:420-500), and its caller verifies every proof it makes (bin/src/main.rs:298), so
// the check has to cost far less than a proof.  This is the optimal ate pairing written for that use:
//   * tower Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u (BN254) / 1 + u
//     (BLS12-381); Karatsuba products, complex squaring, Granger-Scott squaring in the cyclotomic subgroup;
//   * twists: BN254 D-type psi(x, y) = (x w^2, y w^3), BLS12-381 M-type psi(x, y) = (x / w^2, y / w^3); a line through
//     points of the twist with slope lam, evaluated at P = (xP, yP) in G1, is the sparse element
//         yP - lam xP w + (lam xT - yT) w^3          (D-type; slots 0, 3, 4 of the tower)
//         (lam xT - yT) - lam xP w^2 + yP w^3        (M-type, scaled by w^3 which the final power kills; slots 0, 1, 4)
//   * the G2 arguments of a KZG check are the two fixed points of the VerifierKey (h, beta h): the slopes of all
//     doubling / addition steps are computed once per point (affine, one inversion per step) and kept (`prepared`), so
//     a verification performs no G2 arithmetic at all;
//   * Miller loops of one product share the squarings of f; loop scalar 6x + 2 in non-adjacent form plus the two
//     Frobenius steps (BN254), |x| followed by a conjugation (BLS12-381);
//   * final exponentiation: f^(p^6 - 1)(p^2 + 1), then the hard part by the curve's x-chains: BN254 the three
//     exponentiations by x of Fuentes-Castaneda et al., BLS12-381 the multiple 3 (p^4 - p^2 + 1) / r =
//     (x - 1)^2 (x + p)(x^2 + p^2 - 1) + 3 (the factor 3 is prime to r: "is the product one" is unchanged).
// Frobenius constants are powers of xi computed once at first use.  selftest() checks every shortcut against its plain
// definition (Frobenius = power by p, cyclotomic squaring = squaring, the result of the hard part has order r).
#pragma once
#include "hostec.hpp"
#include "hostinv.hpp"

#include <map>
#include <memory>
#include <mutex>
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
    static F dbl(const F& a) { return hostec::hf_add<Q>(a, a); }
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
    static constexpr bool BN = C::ID == 0;
    static constexpr int NQ = Q::N / 2;   // 64-bit limbs of Fq

    // ---- Fq2 ------------------------------------------------------------------------------------------------------
    static E2 z2() { return E2{O::zero(), O::zero()}; }
    static E2 o2() { return E2{O::one(), O::zero()}; }
    static E2 add2(const E2& a, const E2& b) { return E2{O::add(a.c0, b.c0), O::add(a.c1, b.c1)}; }
    static E2 dbl2(const E2& a) { return E2{O::dbl(a.c0), O::dbl(a.c1)}; }
    static E2 sub2(const E2& a, const E2& b) { return E2{O::sub(a.c0, b.c0), O::sub(a.c1, b.c1)}; }
    static E2 neg2(const E2& a) { return E2{O::neg(a.c0), O::neg(a.c1)}; }
    static E2 conj2(const E2& a) { return E2{a.c0, O::neg(a.c1)}; }
    static E2 mul2(const E2& a, const E2& b) {   // (a0 + a1 u)(b0 + b1 u), u^2 = -1 (Karatsuba)
        const F t0 = O::mul(a.c0, b.c0), t1 = O::mul(a.c1, b.c1);
        const F s = O::mul(O::add(a.c0, a.c1), O::add(b.c0, b.c1));
        return E2{O::sub(t0, t1), O::sub(O::sub(s, t0), t1)};
    }
    static E2 sqr2(const E2& a) {   // (a0 + a1)(a0 - a1) + 2 a0 a1 u
        const F m = O::mul(a.c0, a.c1);
        return E2{O::mul(O::add(a.c0, a.c1), O::sub(a.c0, a.c1)), O::dbl(m)};
    }
    static E2 scal2(const E2& a, const F& s) { return E2{O::mul(a.c0, s), O::mul(a.c1, s)}; }
    static E2 mulxi2(const E2& a) {   // times xi = 9 + u (BN254) / 1 + u (BLS12-381), additions only
        if (BN) {
            const F n0 = O::add(O::dbl(O::dbl(O::dbl(a.c0))), a.c0), n1 = O::add(O::dbl(O::dbl(O::dbl(a.c1))), a.c1);
            return E2{O::sub(n0, a.c1), O::add(n1, a.c0)};
        }
        return E2{O::sub(a.c0, a.c1), O::add(a.c0, a.c1)};
    }
    static bool inv2(const E2& a, E2* out) {
        const F d = O::add(O::mul(a.c0, a.c0), O::mul(a.c1, a.c1));
        if (O::is_zero(d)) return false;
        const F di = O::inv(d);
        *out = E2{O::mul(a.c0, di), O::neg(O::mul(a.c1, di))};
        return true;
    }
    static bool is_zero2(const E2& a) { return O::is_zero(a.c0) && O::is_zero(a.c1); }
    static bool eq2(const E2& a, const E2& b) { return O::eq(a.c0, b.c0) && O::eq(a.c1, b.c1); }
    static E2 xi() { return E2{O::from_u32(BN ? 9 : 1), O::one()}; }
    static E2 pow2(const E2& a, const uint64_t* e, int nwords) {
        E2 r = o2();
        bool started = false;
        for (int i = nwords * 64 - 1; i >= 0; --i) {
            if (started) r = sqr2(r);
            if ((e[i / 64] >> (i % 64)) & 1) {
                r = started ? mul2(r, a) : a;
                started = true;
            }
        }
        return r;
    }

    // ---- Fq6 ------------------------------------------------------------------------------------------------------
    static E6 z6() { return E6{z2(), z2(), z2()}; }
    static E6 o6() { return E6{o2(), z2(), z2()}; }
    static E6 add6(const E6& a, const E6& b) { return E6{add2(a.c0, b.c0), add2(a.c1, b.c1), add2(a.c2, b.c2)}; }
    static E6 sub6(const E6& a, const E6& b) { return E6{sub2(a.c0, b.c0), sub2(a.c1, b.c1), sub2(a.c2, b.c2)}; }
    static E6 neg6(const E6& a) { return E6{neg2(a.c0), neg2(a.c1), neg2(a.c2)}; }
    static E6 dbl6(const E6& a) { return E6{dbl2(a.c0), dbl2(a.c1), dbl2(a.c2)}; }
    static E6 mul6(const E6& a, const E6& b) {   // Karatsuba in v, v^3 = xi: six Fq2 products
        const E2 v0 = mul2(a.c0, b.c0), v1 = mul2(a.c1, b.c1), v2 = mul2(a.c2, b.c2);
        const E2 t0 = sub2(sub2(mul2(add2(a.c1, a.c2), add2(b.c1, b.c2)), v1), v2);
        const E2 t1 = sub2(sub2(mul2(add2(a.c0, a.c1), add2(b.c0, b.c1)), v0), v1);
        const E2 t2 = sub2(sub2(mul2(add2(a.c0, a.c2), add2(b.c0, b.c2)), v0), v2);
        return E6{add2(v0, mulxi2(t0)), add2(t1, mulxi2(v2)), add2(t2, v1)};
    }
    static E6 mul6_01(const E6& a, const E2& b0, const E2& b1) {   // times (b0, b1, 0): five Fq2 products
        const E2 v0 = mul2(a.c0, b0), v1 = mul2(a.c1, b1);
        const E2 t1 = sub2(sub2(mul2(add2(a.c0, a.c1), add2(b0, b1)), v0), v1);
        return E6{add2(v0, mulxi2(mul2(a.c2, b1))), t1, add2(mul2(a.c2, b0), v1)};
    }
    static E6 mul6_1(const E6& a, const E2& b1) {   // times (0, b1, 0)
        return E6{mulxi2(mul2(a.c2, b1)), mul2(a.c0, b1), mul2(a.c1, b1)};
    }
    static E6 mulv6(const E6& a) { return E6{mulxi2(a.c2), a.c0, a.c1}; }
    static E6 scalf6(const E6& a, const F& s) { return E6{scal2(a.c0, s), scal2(a.c1, s), scal2(a.c2, s)}; }
    static bool inv6(const E6& a, E6* out) {
        const E2 c0 = sub2(sqr2(a.c0), mulxi2(mul2(a.c1, a.c2)));
        const E2 c1 = sub2(mulxi2(sqr2(a.c2)), mul2(a.c0, a.c1));
        const E2 c2 = sub2(sqr2(a.c1), mul2(a.c0, a.c2));
        const E2 t = add2(mul2(a.c0, c0), mulxi2(add2(mul2(a.c2, c1), mul2(a.c1, c2))));
        E2 ti;
        if (!inv2(t, &ti)) return false;
        *out = E6{mul2(c0, ti), mul2(c1, ti), mul2(c2, ti)};
        return true;
    }

    // ---- Fq12 -----------------------------------------------------------------------------------------------------
    static E12 o12() { return E12{o6(), z6()}; }
    static E12 mul12(const E12& a, const E12& b) {
        const E6 a0b0 = mul6(a.c0, b.c0), a1b1 = mul6(a.c1, b.c1);
        const E6 c1 = sub6(sub6(mul6(add6(a.c0, a.c1), add6(b.c0, b.c1)), a0b0), a1b1);
        return E12{add6(a0b0, mulv6(a1b1)), c1};
    }
    static E12 sqr12(const E12& a) {   // complex squaring: two Fq6 products
        const E6 ab = mul6(a.c0, a.c1);
        const E6 t = mul6(add6(a.c0, a.c1), add6(a.c0, mulv6(a.c1)));
        return E12{sub6(sub6(t, ab), mulv6(ab)), dbl6(ab)};
    }
    static E12 conj12(const E12& a) { return E12{a.c0, neg6(a.c1)}; }
    static bool inv12(const E12& a, E12* out) {
        E6 t;
        if (!inv6(sub6(mul6(a.c0, a.c0), mulv6(mul6(a.c1, a.c1))), &t)) return false;
        *out = E12{mul6(a.c0, t), neg6(mul6(a.c1, t))};
        return true;
    }
    static bool eq12(const E12& a, const E12& b) { return memcmp(&a, &b, sizeof(a)) == 0; }   // limbs are canonical
    static bool is_one12(const E12& a) { return eq12(a, o12()); }
    // a times the sparse line (o0; o3, o4) = o0 + o3 w + o4 w^3 with o0 in Fq  (D-type twist)
    static E12 mul12_034(const E12& a, const F& o0, const E2& o3, const E2& o4) {
        const E6 t0 = scalf6(a.c0, o0);
        const E6 t1 = mul6_01(a.c1, o3, o4);
        const E2 s0{O::add(o3.c0, o0), o3.c1};
        const E6 t2 = mul6_01(add6(a.c0, a.c1), s0, o4);
        return E12{add6(t0, mulv6(t1)), sub6(sub6(t2, t0), t1)};
    }
    // a times the sparse line (o0, o1; o4) = o0 + o1 w^2 + o4 w^3 with o4 in Fq  (M-type twist)
    static E12 mul12_014(const E12& a, const E2& o0, const E2& o1, const F& o4) {
        const E6 t0 = mul6_01(a.c0, o0, o1);
        const E6 t1 = E6{mulxi2(scal2(a.c1.c2, o4)), scal2(a.c1.c0, o4), scal2(a.c1.c1, o4)};   // a.c1 * (0, o4, 0)
        const E2 s1{O::add(o1.c0, o4), o1.c1};
        const E6 t2 = mul6_01(add6(a.c0, a.c1), o0, s1);
        return E12{add6(t0, mulv6(t1)), sub6(sub6(t2, t0), t1)};
    }
    static E12 pow12(const E12& a, const uint64_t* e, int nwords) {
        E12 r = o12();
        bool started = false;
        for (int i = nwords * 64 - 1; i >= 0; --i) {
            if (started) r = sqr12(r);
            if ((e[i / 64] >> (i % 64)) & 1) {
                r = started ? mul12(r, a) : a;
                started = true;
            }
        }
        return r;
    }

    // ---- Frobenius ------------------------------------------------------------------------------------------------
    // a = sum_i a_i w^i (a_i in Fq2; w^0..w^5 = c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2):
    //   a^p = sum conj(a_i) g1[i] w^i,  a^(p^2) = sum a_i g2[i] w^i,  a^(p^3) = sum conj(a_i) g3[i] w^i,
    // g1[i] = xi^(i (p - 1) / 6), g2[i] = g1[i] conj(g1[i]) (in Fq), g3[i] = g1[i] g2[i]
    struct Consts {
        E2 g1[6], g2[6], g3[6];
        uint64_t p[NQ];
    };
    static const Consts& consts() {
        static const Consts k = [] {
            Consts c;
            for (int i = 0; i < NQ; ++i) c.p[i] = hostec::HParams<Q>::mod(i);
            uint64_t e[NQ];   // (p - 1) / 6
            uint64_t rem = 0;
            for (int i = NQ - 1; i >= 0; --i) {
                const hostec::u128 cur = ((hostec::u128)rem << 64) | (c.p[i] - (i == 0 ? 1 : 0));
                e[i] = (uint64_t)(cur / 6);
                rem = (uint64_t)(cur % 6);
            }
            const E2 g = pow2(xi(), e, NQ);
            c.g1[0] = o2();
            for (int i = 1; i < 6; ++i) c.g1[i] = mul2(c.g1[i - 1], g);
            for (int i = 0; i < 6; ++i) {
                c.g2[i] = mul2(c.g1[i], conj2(c.g1[i]));
                c.g3[i] = mul2(c.g1[i], c.g2[i]);
            }
            return c;
        }();
        return k;
    }
    static E12 frob12(const E12& a, int k) {
        const Consts& c = consts();
        const E2* g = k == 1 ? c.g1 : k == 2 ? c.g2 : c.g3;
        const bool cj = (k & 1) != 0;
        auto f = [&](const E2& x, int i) { return mul2(cj ? conj2(x) : x, g[i]); };
        return E12{E6{f(a.c0.c0, 0), f(a.c0.c1, 2), f(a.c0.c2, 4)}, E6{f(a.c1.c0, 1), f(a.c1.c1, 3), f(a.c1.c2, 5)}};
    }

    // Granger-Scott squaring, valid in the cyclotomic subgroup (after the easy part of the final exponentiation)
    static E12 cyclo_sqr12(const E12& a) {
        E2 z0 = a.c0.c0, z4 = a.c0.c1, z3 = a.c0.c2, z2 = a.c1.c0, z1 = a.c1.c1, z5 = a.c1.c2;
        auto fp4_sqr = [](const E2& x, const E2& y, E2* t0, E2* t1) {   // (x + y s)^2, s^2 = xi
            const E2 m = mul2(x, y);
            *t0 = sub2(sub2(mul2(add2(x, y), add2(x, mulxi2(y))), m), mulxi2(m));
            *t1 = dbl2(m);
        };
        E2 t0, t1, t2, t3, t4, t5;
        fp4_sqr(z0, z1, &t0, &t1);
        fp4_sqr(z2, z3, &t2, &t3);
        fp4_sqr(z4, z5, &t4, &t5);
        auto three_minus_two = [](const E2& t, const E2& z) { return add2(dbl2(sub2(t, z)), t); };   // 3 t - 2 z
        auto three_plus_two = [](const E2& t, const E2& z) { return add2(dbl2(add2(t, z)), t); };     // 3 t + 2 z
        z0 = three_minus_two(t0, z0);
        z1 = three_plus_two(t1, z1);
        z2 = three_plus_two(mulxi2(t5), z2);
        z3 = three_minus_two(t4, z3);
        z4 = three_minus_two(t2, z4);
        z5 = three_plus_two(t3, z5);
        return E12{E6{z0, z4, z3}, E6{z2, z1, z5}};
    }

    // the curve parameter |x| (64 bits) and its sign
    static constexpr uint64_t X_ABS = BN ? 4965661367192848881ULL : 0xd201000000010000ULL;
    static constexpr bool X_NEG = !BN;
    static E12 cyclo_pow_xabs(const E12& a) {
        E12 r = a;
        int top = 63;
        while (!((X_ABS >> top) & 1)) --top;
        for (int i = top - 1; i >= 0; --i) {
            r = cyclo_sqr12(r);
            if ((X_ABS >> i) & 1) r = mul12(r, a);
        }
        return r;
    }
    static E12 cyclo_pow_x(const E12& a) {   // a^x for unitary a (inverse = conjugate)
        const E12 r = cyclo_pow_xabs(a);
        return X_NEG ? conj12(r) : r;
    }

    // f^((p^12 - 1) / r) up to a factor prime to r in the exponent; false if f = 0
    static bool final_exponentiation(const E12& f, E12* out) {
        E12 fi;
        if (!inv12(f, &fi)) return false;
        const E12 f1 = mul12(conj12(f), fi);               // f^(p^6 - 1)
        const E12 g = mul12(frob12(f1, 2), f1);             // ^(p^2 + 1): now in the cyclotomic subgroup
        if (BN) {
            // Fuentes-Castaneda, Knapp, Rodriguez-Henriquez: three exponentiations by x (y = g^(-x) chains)
            auto neg_x = [](const E12& a) { return conj12(cyclo_pow_x(a)); };
            const E12 y0 = neg_x(g);
            const E12 y1 = cyclo_sqr12(y0);
            const E12 y2 = cyclo_sqr12(y1);
            E12 y3 = mul12(y2, y1);
            const E12 y4 = neg_x(y3);
            const E12 y5 = cyclo_sqr12(y4);
            E12 y6 = neg_x(y5);
            y3 = conj12(y3);
            y6 = conj12(y6);
            const E12 y7 = mul12(y6, y4);
            const E12 y8 = mul12(y7, y3);
            const E12 y9 = mul12(y8, y1);
            const E12 y10 = mul12(y8, y4);
            const E12 y11 = mul12(y10, g);
            const E12 y13 = mul12(frob12(y9, 1), y11);
            const E12 y14 = mul12(frob12(y8, 2), y13);
            const E12 y15 = frob12(mul12(conj12(g), y9), 3);
            *out = mul12(y15, y14);
            return true;
        }
        // 3 (p^4 - p^2 + 1) / r = (x - 1)^2 (x + p) (x^2 + p^2 - 1) + 3   (Hayashida, Hayasaka, Teruya)
        auto pow_xm1 = [](const E12& a) { return mul12(cyclo_pow_x(a), conj12(a)); };
        const E12 a = pow_xm1(pow_xm1(g));                                    // g^((x - 1)^2)
        const E12 b = mul12(cyclo_pow_x(a), frob12(a, 1));                     // a^(x + p)
        const E12 c = mul12(mul12(cyclo_pow_x(cyclo_pow_x(b)), frob12(b, 2)), conj12(b));   // b^(x^2 + p^2 - 1)
        *out = mul12(c, mul12(cyclo_sqr12(g), g));
        return true;
    }

    struct G1 { F x, y; bool inf; };
    struct G2 { E2 x, y; bool inf; };

    // ---- prepared G2: the slope lam and the intercept mu = lam xT - yT of every line of the Miller loop -----------
    struct Line { E2 lam, mu; };
    struct Prepared {
        bool inf = true;
        std::vector<Line> lines;
    };
    // loop schedule: 'D' doubling step (f is squared first), 'A' addition step
    static const std::vector<char>& schedule() {
        static const std::vector<char> s = [] {
            std::vector<char> ops;
            for (int d : digits()) {
                ops.push_back('D');
                if (d) ops.push_back('A');
            }
            if (BN) {
                ops.push_back('A');   // + pi(Q)
                ops.push_back('A');   // - pi^2(Q)
            }
            return ops;
        }();
        return s;
    }
    // signed digits of the loop scalar below its leading one, most significant first: 6x + 2 in NAF (BN254), |x| (BLS12-381)
    static const std::vector<int>& digits() {
        static const std::vector<int> d = [] {
            std::vector<int> lsb;
            unsigned __int128 k = BN ? (unsigned __int128)6 * X_ABS + 2 : (unsigned __int128)X_ABS;
            while (k) {
                int di = 0;
                if (k & 1) {
                    di = BN ? (((unsigned)(k & 3) == 3) ? -1 : 1) : 1;
                    if (di == 1) k -= 1; else k += 1;
                }
                lsb.push_back(di);
                k >>= 1;
            }
            std::vector<int> out(lsb.rbegin() + 1, lsb.rend());   // drop the leading one
            return out;
        }();
        return d;
    }
    static bool prepare(const G2& Qp, Prepared* out) {
        out->lines.clear();
        out->inf = Qp.inf;
        if (Qp.inf) return true;
        const Consts& k = consts();
        E2 tx = Qp.x, ty = Qp.y;
        const E2 nqy = neg2(Qp.y);
        auto dbl_step = [&]() {
            E2 di;
            if (!inv2(dbl2(ty), &di)) return false;                       // a point of order two: not in G2
            const E2 xx = sqr2(tx);
            const E2 lam = mul2(add2(dbl2(xx), xx), di);
            const E2 nx = sub2(sub2(sqr2(lam), tx), tx);
            out->lines.push_back(Line{lam, sub2(mul2(lam, tx), ty)});
            ty = sub2(mul2(lam, sub2(tx, nx)), ty);
            tx = nx;
            return true;
        };
        auto add_step = [&](const E2& qx, const E2& qy) {
            E2 di;
            if (!inv2(sub2(qx, tx), &di)) return false;                   // T = +-Q inside the loop: Q has small order
            const E2 lam = mul2(sub2(qy, ty), di);
            const E2 nx = sub2(sub2(sqr2(lam), tx), qx);
            out->lines.push_back(Line{lam, sub2(mul2(lam, tx), ty)});
            ty = sub2(mul2(lam, sub2(tx, nx)), ty);
            tx = nx;
            return true;
        };
        for (int d : digits()) {
            if (!dbl_step()) return false;
            if (d == 1 && !add_step(Qp.x, Qp.y)) return false;
            if (d == -1 && !add_step(Qp.x, nqy)) return false;
        }
        if (BN) {
            // pi(Q) = (conj(x) g1[2], conj(y) g1[3]),  -pi^2(Q) = (x g2[2], -y g2[3])   (D-type twist: x ~ w^2, y ~ w^3)
            const E2 q1x = mul2(conj2(Qp.x), k.g1[2]), q1y = mul2(conj2(Qp.y), k.g1[3]);
            const E2 q2x = mul2(Qp.x, k.g2[2]), q2y = neg2(mul2(Qp.y, k.g2[3]));
            if (!add_step(q1x, q1y)) return false;
            if (!add_step(q2x, q2y)) return false;
        }
        return true;
    }

    // prod_i f_{Q_i}(P_i) over prepared Q_i (identities on either side contribute one)
    static E12 miller_product(const G1* ps, const Prepared* const* qs, size_t n) {
        E12 f = o12();
        bool f_is_one = true;
        std::vector<size_t> live;
        std::vector<F> nxp;
        for (size_t i = 0; i < n; ++i)
            if (!ps[i].inf && !qs[i]->inf) {
                live.push_back(i);
                nxp.push_back(O::neg(ps[i].x));
            }
        if (live.empty()) return f;
        const std::vector<char>& ops = schedule();
        for (size_t s = 0; s < ops.size(); ++s) {
            if (ops[s] == 'D' && !f_is_one) f = sqr12(f);
            for (size_t j = 0; j < live.size(); ++j) {
                const size_t i = live[j];
                const Line& l = qs[i]->lines[s];
                const E2 lx = scal2(l.lam, nxp[j]);                            // -lam xP
                f = BN ? mul12_034(f, ps[i].y, lx, l.mu) : mul12_014(f, l.mu, lx, ps[i].y);
            }
            f_is_one = false;
        }
        return X_NEG ? conj12(f) : f;
    }

    // the prepared form of a G2 point is kept: a verifier meets the same two points (h, beta h) in every check
    static std::shared_ptr<const Prepared> prepared_cached(const G2& Qp) {
        static std::mutex mu;
        static std::map<std::vector<uint64_t>, std::shared_ptr<const Prepared>> cache;
        std::vector<uint64_t> key(4 * NQ + 1);
        memcpy(key.data(), &Qp.x, 2 * sizeof(F));
        memcpy(key.data() + 2 * NQ, &Qp.y, 2 * sizeof(F));
        key[4 * NQ] = Qp.inf ? 1 : 0;
        {
            std::lock_guard<std::mutex> lock(mu);
            auto it = cache.find(key);
            if (it != cache.end()) return it->second;
        }
        auto p = std::make_shared<Prepared>();
        if (!prepare(Qp, p.get())) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        if (cache.size() >= 64) cache.clear();   // entries in use stay alive through their shared_ptr
        cache[key] = p;
        return p;
    }

    // prod_i e(P_i, Q_i) == 1 ?   *valid = false when a Q_i cannot be a point of G2 (a zero denominator in its lines)
    static bool product_is_one(const G1* ps, const G2* qs, size_t n, bool* valid = nullptr) {
        std::vector<std::shared_ptr<const Prepared>> keep(n);
        std::vector<const Prepared*> pq(n);
        bool ok = true;
        for (size_t i = 0; i < n; ++i) {
            keep[i] = prepared_cached(qs[i]);
            pq[i] = keep[i].get();
            ok = ok && pq[i] != nullptr;
        }
        if (valid) *valid = ok;
        if (!ok) return false;
        E12 r;
        if (!final_exponentiation(miller_product(ps, pq.data(), n), &r)) return false;
        return is_one12(r);
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
        E2 xi_inv;
        (void)inv2(xi(), &xi_inv);
        const E2 bt = BN ? mul2(b, xi_inv) : mulxi2(b);
        const E2 d = sub2(sqr2(Qt.y), add2(mul2(sqr2(Qt.x), Qt.x), bt));
        return is_zero2(d);
    }

    // every shortcut against its plain definition, on a pseudo-random element; 0 = all good, else the failing check
    static int selftest() {
        E12 a;
        uint64_t s = 0x9E3779B97F4A7C15ULL;
        F* limbs = reinterpret_cast<F*>(&a);
        for (int i = 0; i < 12; ++i) {
            Fe<Q> v = fe_zero<Q>();
            for (int j = 0; j < Q::N - 1; ++j) {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                v.v[j] = (uint32_t)(s >> 32);
            }
            limbs[i] = hostec::hf_from<Q>(fe_to_mont<Q>(v));
        }
        const Consts& k = consts();
        if (!eq12(sqr12(a), mul12(a, a))) return 1;
        E12 ai;
        if (!inv12(a, &ai) || !is_one12(mul12(a, ai))) return 2;
        const E12 ap = pow12(a, k.p, NQ);
        if (!eq12(frob12(a, 1), ap)) return 3;
        if (!eq12(frob12(a, 2), frob12(frob12(a, 1), 1))) return 4;
        if (!eq12(frob12(a, 3), frob12(frob12(a, 2), 1))) return 5;
        // sparse products against the dense one
        const E2 e0 = a.c0.c1, e1 = a.c1.c2;
        const F s0 = a.c0.c0.c1;
        const E12 d034{E6{E2{s0, O::zero()}, z2(), z2()}, E6{e0, e1, z2()}};
        if (!eq12(mul12_034(a, s0, e0, e1), mul12(a, d034))) return 6;
        const E12 d014{E6{e0, e1, z2()}, E6{z2(), E2{s0, O::zero()}, z2()}};
        if (!eq12(mul12_014(a, e0, e1, s0), mul12(a, d014))) return 7;
        // cyclotomic subgroup
        const E12 f1 = mul12(conj12(a), ai);
        const E12 g = mul12(frob12(f1, 2), f1);
        if (!eq12(cyclo_sqr12(g), sqr12(g))) return 8;
        if (!is_one12(mul12(g, conj12(g)))) return 9;
        const uint64_t xa[1] = {X_ABS};
        if (!eq12(cyclo_pow_xabs(g), pow12(g, xa, 1))) return 10;
        // the hard part leaves an element of order r
        E12 h;
        if (!final_exponentiation(a, &h)) return 11;
        typedef typename C::Fr R;
        uint64_t rw[R::N / 2];
        for (int i = 0; i < R::N / 2; ++i) rw[i] = (uint64_t)R::mod(2 * i) | ((uint64_t)R::mod(2 * i + 1) << 32);
        if (is_one12(h) || !is_one12(pow12(h, rw, R::N / 2))) return 12;
        return 0;
    }
};

}  // namespace pairing
}  // namespace zkt
