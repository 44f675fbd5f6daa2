// Carry-free ("unsaturated") field arithmetic for gfx950: 29-bit limbs in 32-bit registers.
//
// Why: on MI355X v_mad_u64_u32 issues every ~4.4 cycles per wave but every carry-propagating
// companion (v_add_co/v_addc 4.3, v_lshl_add_u64 4.2, plus the v_mov pairs the 64-bit operands
// need) costs as much again (profiles/microbench_r01.txt).  With 29-bit limbs a 64-bit accumulator
// absorbs all 2L partial products of a column without overflow, so the Montgomery product is
// 2L^2 + L multiply-adds and nothing else in its inner loops; additions are L independent
// full-rate adds.  L = 9 limbs (261 bits) for the 254/255-bit fields, 14 (406 bits) for Fq381.
//
// Representation: value = sum l[i] * 2^(29 i).  "Normalised": l[i] < 2^29 for i < L-1 (the top
// limb keeps the excess).  Values are lazily reduced: every function states the bound it needs
// and the bound it returns, in multiples of p.  Montgomery radix R' = 2^(29 L) = R * 2^SH where
// R = 2^(32 N) is arkworks' radix; fx_unpack_shift() multiplies by 2^SH for free while unpacking,
// which makes fx_mul(unpack(a~), unpack_shift(b~)) return (ab)~ in arkworks' own Montgomery form.
#pragma once
#include "fp.hpp"
#include <utility>

namespace zkt {

// ---- compile-time big-number helpers (32-bit words, little endian) ---------------------------
template <int N>
struct Words {
    uint32_t w[N];
};

template <int N>
constexpr bool words_geq(const Words<N>& a, const Words<N>& b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a.w[i] > b.w[i]) return true;
        if (a.w[i] < b.w[i]) return false;
    }
    return true;
}
template <int N>
constexpr Words<N> words_sub(const Words<N>& a, const Words<N>& b) {
    Words<N> r{};
    uint64_t borrow = 0;
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)a.w[i] - b.w[i] - borrow;
        r.w[i] = (uint32_t)x;
        borrow = (x >> 63) & 1;
    }
    return r;
}
// (2 * a) mod p for a < p, with one spare word so that the doubling cannot overflow
template <int N>
constexpr Words<N> words_dbl_mod(const Words<N>& a, const Words<N>& p) {
    Words<N> r{};
    uint32_t carry = 0;
    for (int i = 0; i < N; ++i) {
        r.w[i] = (a.w[i] << 1) | carry;
        carry = a.w[i] >> 31;
    }
    // every modulus here has >= 1 spare bit, so 2a < 2^(32N)
    if (words_geq<N>(r, p)) r = words_sub<N>(r, p);
    return r;
}
template <class P>
constexpr Words<P::N> words_modulus() {
    Words<P::N> m{};
    for (int i = 0; i < P::N; ++i) m.w[i] = P::mod(i);
    return m;
}
// 2^k mod p as a canonical integer
template <class P>
constexpr Words<P::N> pow2_mod(int k) {
    Words<P::N> p = words_modulus<P>();
    Words<P::N> r{};
    r.w[0] = 1;
    for (int i = 0; i < k; ++i) r = words_dbl_mod<P::N>(r, p);
    return r;
}

// ---- derived parameters -----------------------------------------------------------------------
template <class P>
struct FxP {
    static constexpr int N = P::N;                       // 32-bit words of the packed form
    static constexpr int L = (32 * P::N + 28) / 29;      // 29-bit limbs: 9 (N = 8), 14 (N = 12)
    static constexpr int SH = 29 * L - 32 * P::N;        // R' = R * 2^SH : 5, 22
    static constexpr uint32_t MASK = (1u << 29) - 1u;
    static constexpr uint32_t INV = P::INV & MASK;       // -p^-1 mod 2^29

    // limb i of a canonical integer given as 32-bit words
    ZKT_HD static constexpr uint32_t limb_of(const Words<P::N>& x, int i) {
        int lo = 29 * i;
        int w = lo >> 5, o = lo & 31;
        uint64_t v = 0;
        if (w < P::N) v = x.w[w];
        if (w + 1 < P::N) v |= (uint64_t)x.w[w + 1] << 32;
        return (uint32_t)(v >> o) & MASK;
    }
    ZKT_HD static constexpr uint32_t mod(int i) {
        constexpr Words<P::N> m = words_modulus<P>();
        return limb_of(m, i);
    }
    // R' mod p (the Montgomery one of this representation), limb i
    ZKT_HD static constexpr uint32_t one(int i) {
        constexpr Words<P::N> v = pow2_mod<P>(29 * L);
        return limb_of(v, i);
    }
    // 2^(32N) mod p = R mod p: fx_mul(x^, this) takes R'-Montgomery x^ to arkworks' R-Montgomery x~
    ZKT_HD static constexpr uint32_t to_ark(int i) {
        constexpr Words<P::N> v = pow2_mod<P>(32 * P::N);
        return limb_of(v, i);
    }
    // 2^(2*29L - 32N) mod p: fx_mul(x~, this) takes arkworks' x~ to x^
    ZKT_HD static constexpr uint32_t from_ark(int i) {
        constexpr Words<P::N> v = pow2_mod<P>(2 * 29 * L - 32 * P::N);
        return limb_of(v, i);
    }
    // limb i of K * p in normalised form (top limb keeps the excess); K <= 64
    ZKT_HD static constexpr uint32_t kmod(int K, int i) {
        uint64_t c = 0;
        uint32_t out = 0;
        for (int j = 0; j <= i; ++j) {
            uint64_t x = (uint64_t)K * mod(j) + c;
            out = (j == L - 1) ? (uint32_t)x : ((uint32_t)x & MASK);
            c = x >> 29;
        }
        return out;
    }
    // limb i of 2^(29 L) - p (normalised): adding q * this instead of subtracting q * p keeps a product chain additive
    ZKT_HD static constexpr uint32_t nmod(int i) {
        uint32_t out = 0;
        uint32_t borrow = 0;
        for (int j = 0; j <= i; ++j) {
            // (2^29 [only conceptually, at the top] - mod(j) - borrow) mod 2^29, limb by limb from the bottom
            const uint32_t m = mod(j) + borrow;
            out = (0u - m) & MASK;
            borrow = m != 0 ? 1u : 0u;
        }
        return out;
    }
    // window used to estimate value / p: bits [OFS, OFS + 32) ; p >> OFS has 24 significant bits
    static constexpr int OFS = P::BITS - 24;
    ZKT_HD static constexpr uint32_t pwin() {
        constexpr Words<P::N> m = words_modulus<P>();
        int w = OFS >> 5, o = OFS & 31;
        uint64_t v = m.w[w];
        if (w + 1 < P::N) v |= (uint64_t)m.w[w + 1] << 32;
        return (uint32_t)(v >> o);
    }
    // floor(2^(32+20) / (pwin + 1)): q = (vwin * QM) >> 52 never overestimates value / p
    ZKT_HD static constexpr uint32_t qm() { return (uint32_t)(((uint64_t)1 << 52) / ((uint64_t)pwin() + 1)); }
};

template <class P>
struct Fx {
    uint32_t l[FxP<P>::L];
};

template <class P>
ZKT_HD Fx<P> fx_zero() {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = 0;
    return r;
}
template <class P>
ZKT_HD Fx<P> fx_one() {  // R' mod p
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = FxP<P>::one(i);
    return r;
}
template <class P>
ZKT_HD Fx<P> fx_const_to_ark() {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = FxP<P>::to_ark(i);
    return r;
}
template <class P>
ZKT_HD Fx<P> fx_const_from_ark() {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = FxP<P>::from_ark(i);
    return r;
}

// ---- packed (8 / 12 x u32) <-> limbs ---------------------------------------------------------------
// value(a) * 2^S, S in {0, SH}: limb i = bits [29 i - S, 29 i - S + 29) of a.  Output normalised, < 2^S * a.
template <class P, int S>
ZKT_HD Fx<P> fx_unpack_s(const Fe<P>& a) {
    constexpr int L = FxP<P>::L, N = P::N;
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int lo = 29 * i - S;  // may be negative for i = 0
        uint32_t v;
        if (lo < 0) {
            v = a.v[0] << (-lo);
        } else {
            const int w = lo >> 5, o = lo & 31;
            uint32_t x0 = (w < N) ? a.v[w] : 0u;
            uint32_t x1 = (w + 1 < N) ? a.v[w + 1] : 0u;
            v = o ? ((x0 >> o) | (x1 << (32 - o))) : x0;
        }
        r.l[i] = v & FxP<P>::MASK;  // nothing lives above bit 29 L - 1 either
    }
    return r;
}
template <class P>
ZKT_HD Fx<P> fx_unpack(const Fe<P>& a) { return fx_unpack_s<P, 0>(a); }
template <class P>
ZKT_HD Fx<P> fx_unpack_shift(const Fe<P>& a) { return fx_unpack_s<P, FxP<P>::SH>(a); }

// normalised limbs of a value < 2^(32N) -> packed words
template <class P>
ZKT_HD Fe<P> fx_pack(const Fx<P>& a) {
    constexpr int L = FxP<P>::L, N = P::N;
    Fe<P> r;
#pragma unroll
    for (int w = 0; w < N; ++w) {
        const int lo = 32 * w;
        const int i = lo / 29, o = lo - 29 * i;   // word w starts at bit o of limb i
        uint32_t v = a.l[i] >> o;                  // 29 - o bits
        if (i + 1 < L) v |= a.l[i + 1] << (29 - o);
        if (29 - o + 29 < 32 && i + 2 < L) v |= a.l[i + 2] << (58 - o);
        r.v[w] = v;
    }
    return r;
}

// ---- carries ---------------------------------------------------------------------------------------
// limbs may hold up to 32 bits; afterwards l[i] < 2^29 for i < L-1.  Value unchanged.
template <class P>
ZKT_HD Fx<P> fx_normalize(const Fx<P>& a) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        uint32_t x = a.l[i] + c;   // callers keep limbs < 2^32 - 2^4
        r.l[i] = x & FxP<P>::MASK;
        c = x >> 29;
    }
    r.l[L - 1] = a.l[L - 1] + c;
    return r;
}

// a + b, limbwise; result normalised, value = a + b (must stay < 2^(29 L))
template <class P>
ZKT_HD Fx<P> fx_add(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        uint32_t x = a.l[i] + b.l[i] + c;
        r.l[i] = x & FxP<P>::MASK;
        c = x >> 29;
    }
    r.l[L - 1] = a.l[L - 1] + b.l[L - 1] + c;
    return r;
}

// a + K p - b, needs b <= K p (as values); result normalised, value < a + K p
template <class P, int K>
ZKT_HD Fx<P> fx_sub(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        // a.l, b.l < 2^29 (+ small), kmod < 2^29: everything fits a signed 32-bit limb
        int32_t x = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)FxP<P>::kmod(K, i) + c;
        r.l[i] = (uint32_t)x & FxP<P>::MASK;
        c = x >> 29;
    }
    int32_t x = (int32_t)a.l[L - 1] - (int32_t)b.l[L - 1] + (int32_t)FxP<P>::kmod(K, L - 1) + c;
    r.l[L - 1] = (uint32_t)x;
    return r;
}

// a + K p - b with no carry propagation, for values that go straight into a product.  K p is re-expressed with
// 2^29 borrowed into every limb from the one above, so each limb difference is non-negative whenever b's limbs are
// normalised and b <= (K - 1) p (that keeps the top limb non-negative too).  Result limbs < 2^29 + 2^30: legal as
// ONE operand of fx_mul / fx_mul_inl when the other is normalised (column sums stay below 2^64).
template <class P, int K>
ZKT_HD Fx<P> fx_sub_lazy(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t fat = FxP<P>::kmod(K, i) + (i < L - 1 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
        r.l[i] = a.l[i] + (fat - b.l[i]);
    }
    return r;
}

// a + b limb by limb, no carry pass: for sums that go through at most two such steps before a product or fx_reduce_lazy
// brings the limbs back below 2^29 (limbs stay below 2^31)
template <class P>
ZKT_HD Fx<P> fx_add_lazy(const Fx<P>& a, const Fx<P>& b) {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// fx_sub_lazy for a subtrahend whose limbs may reach 2^BB (a lazy sum: BB = 30): K p is re-expressed with 2^BB borrowed
// into every limb.  Needs b <= (K - 1) p and b's limbs <= 2^BB; result limbs < a's + 2^BB + 2^29.
template <class P, int K, int BB>
ZKT_HD Fx<P> fx_sub_lazy_wide(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    static_assert(BB >= 29 && BB <= 30, "borrow of one or two limb widths");
    constexpr uint32_t up = 1u << (BB - 29);   // what the limb above gives up
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t fat = FxP<P>::kmod(K, i) + (i < L - 1 ? (1u << BB) : 0u) - (i > 0 ? up : 0u);
        r.l[i] = a.l[i] + (fat - b.l[i]);
    }
    return r;
}

// a + K p - b - 2c in one carry pass (b + 2c <= K p); result normalised, value < a + K p
template <class P, int K>
ZKT_HD Fx<P> fx_sub2(const Fx<P>& a, const Fx<P>& b, const Fx<P>& c2) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        // |x| < 2^29 + 2^29 + 2^30 + 2^29: fits a signed 32-bit limb
        int32_t x = (int32_t)a.l[i] - (int32_t)b.l[i] - (int32_t)(c2.l[i] << 1) + (int32_t)FxP<P>::kmod(K, i) + c;
        r.l[i] = (uint32_t)x & FxP<P>::MASK;
        c = x >> 29;
    }
    int32_t x = (int32_t)a.l[L - 1] - (int32_t)b.l[L - 1] - (int32_t)(c2.l[L - 1] << 1) + (int32_t)FxP<P>::kmod(K, L - 1) + c;
    r.l[L - 1] = (uint32_t)x;
    return r;
}

template <class P>
ZKT_HD Fx<P> fx_dbl(const Fx<P>& a) { return fx_add<P>(a, a); }

// ---- Montgomery product ------------------------------------------------------------------------------
// a * b / R' mod p.  Needs normalised limbs (a.l, b.l < 2^29, top limb free) and a * b < R' * p as values
// (e.g. a < 8p, b < 8p).  Returns normalised limbs, value < 2p.
// The body is a real (non-inlined) function so that one copy stays hot in the instruction cache; its
// operands travel as L-element vectors because a 9-word struct would be returned through scratch memory.
template <class P>
struct FxVec {
    typedef uint32_t type __attribute__((ext_vector_type(FxP<P>::L)));
};

// The device products below issue their multiply-adds through opaque statements: left to itself the compiler
// starts every column from zero and then ADDS the shifted carry of the previous one (a v_lshl_add_u64 per column, 7 %
// of the MSM accumulation loop); v_mad_u64_u32 takes the carry as its addend for free when the chain is kept in order.
// One operand may be a scalar register (the modulus limbs are compile-time constants).
#if defined(__HIP_DEVICE_COMPILE__)
#include "fx_madrow.inc"
__device__ __forceinline__ uint64_t fx_mad0(uint32_t a, uint32_t b) {
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
}
template <int N, bool K>
__device__ __forceinline__ void fx_mad_rows(uint64_t& acc, const uint32_t* x, const uint32_t* y) {
    if constexpr (N > 14) {
        fx_mad_rows<14, K>(acc, x, y);
        fx_mad_rows<N - 14, K>(acc, x + 14, y + 14);
    } else if constexpr (N > 0) {
        if constexpr (K) MadRow<N>::vk(acc, x, y);
        else MadRow<N>::vv(acc, x, y);
    }
}

// n multiply-adds of a column whose length is known only after the caller's loops are unrolled (n <= L <= 14)
template <class P>
__device__ __forceinline__ void fx_mad_rows_dyn(uint64_t& acc, const uint32_t* x, const uint32_t* y, int n) {
    switch (n) {
        case 0: break;
        case 1: MadRow<1>::vv(acc, x, y); break;
        case 2: MadRow<2>::vv(acc, x, y); break;
        case 3: MadRow<3>::vv(acc, x, y); break;
        case 4: MadRow<4>::vv(acc, x, y); break;
        case 5: MadRow<5>::vv(acc, x, y); break;
        case 6: MadRow<6>::vv(acc, x, y); break;
        case 7: MadRow<7>::vv(acc, x, y); break;
        case 8: MadRow<8>::vv(acc, x, y); break;
        case 9: MadRow<9>::vv(acc, x, y); break;
        case 10: MadRow<10>::vv(acc, x, y); break;
        case 11: MadRow<11>::vv(acc, x, y); break;
        case 12: MadRow<12>::vv(acc, x, y); break;
        case 13: MadRow<13>::vv(acc, x, y); break;
        default: MadRow<14>::vv(acc, x, y); break;
    }
}
template <class P>
__device__ __forceinline__ void fx_mad_rows_k(uint64_t& acc, const uint32_t* x, const uint32_t* y, int n) {   // y: constants
    switch (n) {
        case 0: break;
        case 1: MadRow<1>::vk(acc, x, y); break;
        case 2: MadRow<2>::vk(acc, x, y); break;
        case 3: MadRow<3>::vk(acc, x, y); break;
        case 4: MadRow<4>::vk(acc, x, y); break;
        case 5: MadRow<5>::vk(acc, x, y); break;
        case 6: MadRow<6>::vk(acc, x, y); break;
        case 7: MadRow<7>::vk(acc, x, y); break;
        case 8: MadRow<8>::vk(acc, x, y); break;
        case 9: MadRow<9>::vk(acc, x, y); break;
        case 10: MadRow<10>::vk(acc, x, y); break;
        case 11: MadRow<11>::vk(acc, x, y); break;
        case 12: MadRow<12>::vk(acc, x, y); break;
        case 13: MadRow<13>::vk(acc, x, y); break;
        default: MadRow<14>::vk(acc, x, y); break;
    }
}

// Column COL of (a * b [+ c * d]) / R' in product-scanning order (SQR: b == a, cross terms once against the doubled
// operand a2): the operand products as one chain, the reduction products m[i] * p[j] as another.
template <class P, bool TWO, bool SQR, int COL>
__device__ __forceinline__ void fx_mont_column(uint64_t& acc, uint32_t* m, Fx<P>& r, const uint32_t* a, const uint32_t* b,
                                               const uint32_t* c, const uint32_t* d, const uint32_t* a2) {
    constexpr int L = FxP<P>::L;
    constexpr int lo = COL < L ? 0 : COL - L + 1, hi = COL < L ? COL : L - 1;     // i + j = COL, both below L
    constexpr int terms = hi - lo + 1;
    constexpr int nv = SQR ? (terms + 1) / 2 : (TWO ? 2 * terms : terms);
    constexpr int skip = COL == 0 ? 1 : 0;                                         // a0 * b0 started the accumulator
    uint32_t x[nv + 1], y[nv + 1];
    int n = 0;
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
        const int j = COL - i;
        if (SQR) {
            if (j > i) { x[n] = a2[i]; y[n] = a[j]; ++n; }
            else if (j == i) { x[n] = a[i]; y[n] = a[i]; ++n; }
        } else {
            x[n] = a[i]; y[n] = b[j]; ++n;
            if (TWO) { x[n] = c[i]; y[n] = d[j]; ++n; }
        }
    }
    fx_mad_rows<nv - skip, false>(acc, x + skip, y + skip);
    constexpr int rlo = lo, rhi = COL < L ? COL - 1 : L - 1;                        // m[i] * p[COL - i], COL - i >= 1
    constexpr int nk = rhi - rlo + 1;
    if constexpr (nk > 0) {
        uint32_t xm[nk], yk[nk];
#pragma unroll
        for (int t = 0; t < nk; ++t) {
            xm[t] = m[rlo + t];
            yk[t] = FxP<P>::mod(COL - rlo - t);
        }
        fx_mad_rows<nk, true>(acc, xm, yk);
    }
    if constexpr (COL < L) {
        m[COL] = ((uint32_t)acc * FxP<P>::INV) & FxP<P>::MASK;
        const uint32_t p0 = FxP<P>::mod(0);
        MadRow<1>::vk(acc, &m[COL], &p0);
    } else {
        r.l[COL - L] = (uint32_t)acc & FxP<P>::MASK;
    }
    acc >>= 29;
}

template <class P, bool TWO, bool SQR, int... COLS>
__device__ __forceinline__ Fx<P> fx_mont_chain_seq(const Fx<P>& a, const Fx<P>& b, const Fx<P>& c, const Fx<P>& d,
                                                   std::integer_sequence<int, COLS...>) {
    constexpr int L = FxP<P>::L;
    uint32_t m[L], a2[L];
    if (SQR) {
#pragma unroll
        for (int i = 0; i < L; ++i) a2[i] = a.l[i] << 1;
    }
    Fx<P> r;
    uint64_t acc = fx_mad0(a.l[0], SQR ? a.l[0] : b.l[0]);
    (fx_mont_column<P, TWO, SQR, COLS>(acc, m, r, a.l, b.l, c.l, d.l, a2), ...);
    r.l[L - 1] = (uint32_t)acc;
    return r;
}
template <class P, bool TWO, bool SQR>
__device__ __forceinline__ Fx<P> fx_mont_chain(const Fx<P>& a, const Fx<P>& b, const Fx<P>& c, const Fx<P>& d) {
    return fx_mont_chain_seq<P, TWO, SQR>(a, b, c, d, std::make_integer_sequence<int, 2 * FxP<P>::L - 1>());
}
#endif

// fully inlined product (for the one or two loops that are hot enough to own a private copy)
template <class P>
ZKT_HD Fx<P> fx_mul_inl(const Fx<P>& a, const Fx<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fx_mont_chain<P, false, false>(a, b, a, b);
#else
    constexpr int L = FxP<P>::L;
    uint64_t t[L + 1];
#pragma unroll
    for (int j = 0; j <= L; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int j = 0; j < L; ++j) t[j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t m = ((uint32_t)t[0] * FxP<P>::INV) & FxP<P>::MASK;
#pragma unroll
        for (int j = 0; j < L; ++j) t[j] += (uint64_t)m * FxP<P>::mod(j);
        const uint64_t carry = t[0] >> 29;  // low 29 bits are zero now
#pragma unroll
        for (int j = 0; j < L; ++j) t[j] = t[j + 1];
        t[0] += carry;
        t[L] = 0;
    }
    Fx<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < L - 1; ++j) {
        uint64_t x = t[j] + c;
        r.l[j] = (uint32_t)x & FxP<P>::MASK;
        c = x >> 29;
    }
    r.l[L - 1] = (uint32_t)(t[L - 1] + c);
    return r;
#endif
}

// a^2 / R' with the cross products taken once against the doubled operand: L(L+1)/2 + L^2 multiply-adds.
// Same contract as fx_mul_inl (a * a < R' * p, limbs < 2^29).
template <class P>
ZKT_HD Fx<P> fx_sqr_inl(const Fx<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fx_mont_chain<P, false, true>(a, a, a, a);
#else
    constexpr int L = FxP<P>::L;
    uint32_t a2[L], m[L];
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = a.l[i] << 1;
    Fx<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int j = k - i;
            if (j > i && j < L) acc += (uint64_t)a2[i] * a.l[j];
        }
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int j = k - i;
            if (j >= 0 && j < L && i < k && i < L) {
                if (k < L || i > k - L) acc += (uint64_t)m[i] * FxP<P>::mod(j);
            }
        }
        if (k < L) {
            m[k] = ((uint32_t)acc * FxP<P>::INV) & FxP<P>::MASK;
            acc += (uint64_t)m[k] * FxP<P>::mod(0);
        } else {
            r.l[k - L] = (uint32_t)acc & FxP<P>::MASK;
        }
        acc >>= 29;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
#endif
}

// (a * b + c * d) / R' with one reduction: 3 L^2 multiply-adds instead of 4 L^2.
// Needs a * b + c * d < R' * p and limbs < 2^29; returns normalised limbs, value < 2p.
template <class P>
ZKT_HD Fx<P> fx_mul2_inl(const Fx<P>& a, const Fx<P>& b, const Fx<P>& c, const Fx<P>& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fx_mont_chain<P, true, false>(a, b, c, d);
#else
    constexpr int L = FxP<P>::L;
    uint32_t m[L];
    Fx<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int j = k - i;
            if (j >= 0 && j < L) {
                acc += (uint64_t)a.l[i] * b.l[j];
                acc += (uint64_t)c.l[i] * d.l[j];
                if (i < k && (k < L || i > k - L)) acc += (uint64_t)m[i] * FxP<P>::mod(j);
            }
        }
        if (k < L) {
            m[k] = ((uint32_t)acc * FxP<P>::INV) & FxP<P>::MASK;
            acc += (uint64_t)m[k] * FxP<P>::mod(0);
        } else {
            r.l[k - L] = (uint32_t)acc & FxP<P>::MASK;
        }
        acc >>= 29;
    }
    r.l[L - 1] = (uint32_t)acc;
    return r;
#endif
}

template <class P>
ZKT_MUL typename FxVec<P>::type fx_mul_raw(typename FxVec<P>::type a, typename FxVec<P>::type b) {
    constexpr int L = FxP<P>::L;
    Fx<P> x, y;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    const Fx<P> z = fx_mul_inl<P>(x, y);
    typename FxVec<P>::type r;
#pragma unroll
    for (int i = 0; i < L; ++i) r[i] = z.l[i];
    return r;
}

template <class P>
ZKT_HD Fx<P> fx_mul(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    typename FxVec<P>::type va, vb;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        va[i] = a.l[i];
        vb[i] = b.l[i];
    }
    typename FxVec<P>::type vr = fx_mul_raw<P>(va, vb);
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.l[i] = vr[i];
    return r;
}
template <class P>
ZKT_HD Fx<P> fx_sqr(const Fx<P>& a) { return fx_mul<P>(a, a); }

// value < 2^6 p (normalised) -> value' = value - q p in [0, 2p), same residue
template <class P>
ZKT_HD Fx<P> fx_reduce_small(const Fx<P>& a) {
    constexpr int L = FxP<P>::L;
    constexpr int OFS = FxP<P>::OFS;
    constexpr int i0 = OFS / 29, o = OFS - 29 * i0;
    // 32-bit window of the value starting at bit OFS (value < 2^(BITS + 6) -> window < 2^30)
    uint64_t win = (uint64_t)a.l[i0] >> o;
    if (i0 + 1 < L) win |= (uint64_t)a.l[i0 + 1] << (29 - o);
    if (i0 + 2 < L) win |= (uint64_t)a.l[i0 + 2] << (58 - o);
    const uint32_t q = (uint32_t)(((uint64_t)(uint32_t)win * FxP<P>::qm()) >> 52);
    Fx<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        int64_t x = (int64_t)a.l[i] - (int64_t)((uint64_t)q * FxP<P>::mod(i)) + c;
        r.l[i] = (uint32_t)x & FxP<P>::MASK;
        c = x >> 29;
    }
    int64_t x = (int64_t)a.l[L - 1] - (int64_t)((uint64_t)q * FxP<P>::mod(L - 1)) + c;
    r.l[L - 1] = (uint32_t)x;
    return r;
}

// value < 2p (normalised) -> canonical [0, p)
template <class P>
ZKT_HD Fx<P> fx_cond_sub_p(const Fx<P>& a) {
    constexpr int L = FxP<P>::L;
    Fx<P> d;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        int32_t x = (int32_t)a.l[i] - (int32_t)FxP<P>::mod(i) + c;
        d.l[i] = (uint32_t)x & FxP<P>::MASK;
        c = x >> 29;
    }
    int32_t top = (int32_t)a.l[L - 1] - (int32_t)FxP<P>::mod(L - 1) + c;
    d.l[L - 1] = (uint32_t)top;
    const bool neg = top < 0;
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}
// value < 2^6 p -> canonical
template <class P>
ZKT_HD Fx<P> fx_canon(const Fx<P>& a) { return fx_cond_sub_p<P>(fx_reduce_small<P>(a)); }

template <class P>
ZKT_HD bool fx_is_zero_canon(const Fx<P>& a) {  // a canonical
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) o |= a.l[i];
    return o == 0;
}

// ---- product by a constant with a precomputed quotient (Shoup / Barrett with a per-constant reciprocal) -----------
// For a fixed w < p let wq = floor(w * 2^(29 L) / p) (L limbs).  Then for any x < 2^(29 L)
//     q = floor(x * wq / 2^(29 L))  is  floor(x w / p)  or one less,   and   x w - q p  is in [0, 2p).
// q needs only the high half of x * wq: columns 7 .. 2L-2 here (dropping the seven lowest columns changes q by at most
// one more), 53 multiply-adds for L = 9; r = x w - q p needs only the low L limbs of x w + q (2^(29 L) - p): 90.  143
// multiply-adds and no m-chain (v_mul_lo + mask per limb) against the Montgomery product's 171: the NTT's butterfly
// twiddles (tables in LDS, both words per entry) use it; one-off products and memory-resident tables stay Montgomery.
// Input: limbs <= 2^31.3 (a lazy sum or difference), value < 2^(29 L); w, wq normalised.  Output normalised, < 3p.
// No Montgomery factor is involved: x w mod p in whatever form x is in, w a plain canonical integer.
template <class P>
ZKT_HD Fx<P> fx_mul_shoup(const Fx<P>& x, const Fx<P>& w, const Fx<P>& wq) {
    constexpr int L = FxP<P>::L;
    constexpr int K0 = L - 2;          // first column of x * wq that is kept
    uint32_t q[L];
#if defined(__HIP_DEVICE_COMPILE__)
    {
        uint64_t acc = 0;
        bool first = true;
#pragma unroll
        for (int k = K0; k <= 2 * L - 2; ++k) {
            const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
            uint32_t xs[L], ys[L];
            int n = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) { xs[n] = x.l[i]; ys[n] = wq.l[k - i]; ++n; }
            if (first) {
                acc = fx_mad0(xs[0], ys[0]);
                fx_mad_rows_dyn<P>(acc, xs + 1, ys + 1, n - 1);
                first = false;
            } else {
                fx_mad_rows_dyn<P>(acc, xs, ys, n);
            }
            if (k >= L) q[k - L] = (uint32_t)acc & FxP<P>::MASK;
            acc >>= 29;
        }
        q[L - 1] = (uint32_t)acc;
    }
    Fx<P> r;
    {
        uint64_t acc = fx_mad0(x.l[0], w.l[0]);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            uint32_t xs[L], ys[L], qs[L], ks[L];
            int n = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) { xs[n] = x.l[i]; ys[n] = w.l[k - i]; qs[n] = q[i]; ks[n] = FxP<P>::nmod(k - i); ++n; }
            if (k == 0) {
                fx_mad_rows_k<P>(acc, qs, ks, 1);
            } else {
                fx_mad_rows_dyn<P>(acc, xs, ys, n);
                fx_mad_rows_k<P>(acc, qs, ks, n);
            }
            r.l[k] = (uint32_t)acc & FxP<P>::MASK;
            acc >>= 29;
        }
    }
    return r;
#else
    {
        uint64_t acc = 0;
        for (int k = K0; k <= 2 * L - 2; ++k) {
            for (int i = 0; i < L; ++i) {
                const int j = k - i;
                if (j >= 0 && j < L) acc += (uint64_t)x.l[i] * wq.l[j];
            }
            if (k >= L) q[k - L] = (uint32_t)acc & FxP<P>::MASK;
            acc >>= 29;
        }
        q[L - 1] = (uint32_t)acc;
    }
    Fx<P> r;
    uint64_t acc = 0;
    for (int k = 0; k < L; ++k) {
        for (int i = 0; i <= k; ++i) {
            acc += (uint64_t)x.l[i] * w.l[k - i];
            acc += (uint64_t)q[i] * FxP<P>::nmod(k - i);
        }
        r.l[k] = (uint32_t)acc & FxP<P>::MASK;
        acc >>= 29;
    }
    return r;
#endif
}

// low L limbs of a * b (normalised operands): wq = low(w^ * (-p^-1 mod 2^(29 L))) for w^ = w 2^(29 L) mod p (table setup)
template <class P>
ZKT_HD Fx<P> fx_mul_low(const Fx<P>& a, const Fx<P>& b) {
    constexpr int L = FxP<P>::L;
    Fx<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
        r.l[k] = (uint32_t)acc & FxP<P>::MASK;
        acc >>= 29;
    }
    return r;
}

// value < 2^6 p with limbs up to 2^31 (lazy sums) -> normalised, < 3p, same residue: the quotient estimate comes from
// the two top limbs (a carry pending below them can only make it smaller), the subtraction of q p is the carry pass
template <class P>
ZKT_HD Fx<P> fx_reduce_lazy(const Fx<P>& a) {
    constexpr int L = FxP<P>::L;
    // value >= top * 2^(29 (L-1)) with top = l[L-1] + (l[L-2] >> 29); p < (ptop + 1) * 2^(29 (L-1))
    constexpr uint32_t ptop = FxP<P>::mod(L - 1);
    static_assert(ptop >= (1u << 16), "the top limb of p must carry the quotient estimate");
    constexpr uint32_t rec = (uint32_t)(((uint64_t)1 << 40) / ((uint64_t)ptop + 1));   // floor(2^40 / (ptop + 1))
    const uint32_t top = a.l[L - 1] + (a.l[L - 2] >> 29);        // < 2^6 * 2^(BITS - 29 (L-1)) + 4: well below 2^32
    const uint32_t q = (uint32_t)(((uint64_t)top * rec) >> 40);  // <= top / (ptop + 1) <= value / p
    Fx<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
        int64_t v = (int64_t)a.l[i] - (int64_t)((uint64_t)q * FxP<P>::mod(i)) + c;
        r.l[i] = (uint32_t)v & FxP<P>::MASK;
        c = v >> 29;
    }
    int64_t v = (int64_t)a.l[L - 1] - (int64_t)((uint64_t)q * FxP<P>::mod(L - 1)) + c;
    r.l[L - 1] = (uint32_t)v;
    return r;
}

// -p^-1 mod 2^(29 L) as L limbs (Hensel lifting on 32-bit words; host, table setup only)
template <class P>
inline Fx<P> fx_neg_p_inverse() {
    constexpr int L = FxP<P>::L;
    constexpr int W = (29 * L + 31) / 32;
    auto mul_lo = [=](const uint32_t* a, const uint32_t* b, uint32_t* r) {
        uint32_t out[W];
        uint64_t carry = 0;
        for (int k = 0; k < W; ++k) {   // column k, accumulated in two halves to stay inside 64 bits
            uint64_t lo = carry & 0xffffffffull, hi = carry >> 32;
            for (int i = 0; i <= k; ++i) {
                const uint64_t pr = (uint64_t)a[i] * b[k - i];
                lo += pr & 0xffffffffull;
                hi += pr >> 32;
            }
            hi += lo >> 32;
            out[k] = (uint32_t)lo;
            carry = hi;
        }
        for (int k = 0; k < W; ++k) r[k] = out[k];
    };
    uint32_t p[W] = {}, x[W] = {};
    for (int i = 0; i < P::N; ++i) p[i] = P::mod(i);
    x[0] = 1;   // p x = 1 mod 2
    for (int it = 0; it < 10; ++it) {   // x <- x (2 - p x): the number of correct bits doubles
        uint32_t t[W], u[W];
        mul_lo(p, x, t);
        uint32_t borrow = 0;
        for (int k = 0; k < W; ++k) {   // u = 2 - t
            const uint64_t v = (uint64_t)(k == 0 ? 2u : 0u) - t[k] - borrow;
            u[k] = (uint32_t)v;
            borrow = (uint32_t)(v >> 63);
        }
        mul_lo(x, u, t);
        for (int k = 0; k < W; ++k) x[k] = t[k];
    }
    uint32_t neg[W + 1] = {};
    uint32_t borrow = 0;
    for (int k = 0; k < W; ++k) {
        const uint64_t v = (uint64_t)0 - x[k] - borrow;
        neg[k] = (uint32_t)v;
        borrow = (uint32_t)(v >> 63);
    }
    Fx<P> r;
    for (int i = 0; i < L; ++i) {
        const int lo = 29 * i, w = lo >> 5, o = lo & 31;
        const uint64_t v = (uint64_t)neg[w] | ((uint64_t)neg[w + 1] << 32);
        r.l[i] = (uint32_t)(v >> o) & FxP<P>::MASK;
    }
    return r;
}

// ---- arkworks-form product on packed operands: the drop-in behind fe_mul ---------------------------
// a~ * b~ -> (ab)~, all canonical packed, R = 2^(32N) Montgomery form on both sides.
template <class P>
ZKT_HD Fe<P> fe_mul_via_fx(const Fe<P>& a, const Fe<P>& b) {
    Fx<P> r = fx_mul<P>(fx_unpack<P>(a), fx_unpack_shift<P>(b));  // < p * 2^SH * p / R' + p < 2p
    return fx_pack<P>(fx_cond_sub_p<P>(r));
}
template <class P>
__host__ __device__ __forceinline__ Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    return fe_mul_via_fx<P>(a, b);
}

// arkworks packed (R form, canonical) <-> R'-Montgomery limbs
template <class P>
ZKT_HD Fx<P> fx_from_ark(const Fe<P>& a) {  // x~ -> x^ (< 2p)
    return fx_mul<P>(fx_unpack<P>(a), fx_const_from_ark<P>());
}
template <class P>
ZKT_HD Fe<P> fx_to_ark(const Fx<P>& a) {    // x^ (< 8p) -> canonical packed x~
    return fx_pack<P>(fx_cond_sub_p<P>(fx_mul<P>(a, fx_const_to_ark<P>())));
}

}  // namespace zkt
