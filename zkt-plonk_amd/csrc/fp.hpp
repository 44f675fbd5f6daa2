// Prime-field arithmetic for gfx950 (CDNA4): 32-bit limbs, Montgomery form, R = 2^(32*N).
//
// Memory format is arkworks' in-memory format (ark-ff 0.3 Fp256/Fp384: little-endian u64 limbs of
// a*R mod p), so vectors cross the C-ABI without repacking: a u64 limb is two consecutive u32 limbs.
// The hot multiply is written so that hipcc lowers every partial product to one v_mad_u64_u32
// (32x32+64 -> 64); there is no dense contraction here, so MFMA is deliberately unused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZKT_HD __host__ __device__ __forceinline__
#define ZKT_D __device__ __forceinline__
// The Montgomery product is ~550 instructions; it is a real function (one copy per field stays hot in
// the instruction cache) instead of being inlined dozens of times into every kernel.
#define ZKT_MUL __host__ __device__ __attribute__((noinline))

namespace zkt {

// ---- parameter packs -------------------------------------------------------------------------
// MOD: modulus limbs; INV = -p^-1 mod 2^32; ONE = R mod p; R2 = R^2 mod p.
// Values are those of ark-bn254 / ark-bls12-381 0.3 (checked against big integers in
// tests/test_constants.py through zkt_debug_params()).
struct Bn254Fr {
    static constexpr int N = 8;
    static constexpr int BITS = 254;
    ZKT_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                   0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                   0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr int TWO_ADICITY = 28;
    static constexpr uint32_t GENERATOR = 5;
};

struct Bls381Fr {
    static constexpr int N = 8;
    static constexpr int BITS = 255;
    ZKT_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t m[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xffffffffu;
    static constexpr int TWO_ADICITY = 32;
    static constexpr uint32_t GENERATOR = 7;
};

struct Bn254Fq {
    static constexpr int N = 8;
    static constexpr int BITS = 254;
    ZKT_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                   0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xe4866389u;
};

struct Bls381Fq {
    static constexpr int N = 12;
    static constexpr int BITS = 381;
    ZKT_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                    0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                    0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t m[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                    0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                    0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return m[i];
    }
    ZKT_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                    0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                    0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xfffcfffdu;
};

// ---- field element ---------------------------------------------------------------------------
template <class P>
struct alignas(16) Fe {
    uint32_t v[P::N];
};

template <class P>
ZKT_HD Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = 0;
    return r;
}
template <class P>
ZKT_HD Fe<P> fe_one() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = P::one(i);
    return r;
}
template <class P>
ZKT_HD Fe<P> fe_r2() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = P::r2(i);
    return r;
}
template <class P>
ZKT_HD bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) o |= a.v[i];
    return o == 0;
}
template <class P>
ZKT_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = a - p if a >= p (a < 2p on entry)
template <class P>
ZKT_HD Fe<P> fe_reduce_once(const Fe<P>& a) {
    Fe<P> d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t x = (uint64_t)a.v[i] - P::mod(i) - borrow;
        d.v[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

template <class P>
ZKT_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> s;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t x = (uint64_t)a.v[i] + b.v[i] + carry;
        s.v[i] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    // every modulus here leaves >= 1 spare bit in the top limb, so a + b < 2^(32N): no carry out
    return fe_reduce_once<P>(s);
}

template <class P>
ZKT_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - borrow;
        d.v[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    uint32_t carry = 0;
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint64_t x = (uint64_t)d.v[i] + (borrow ? P::mod(i) : 0u) + carry;
        r.v[i] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    return r;
}

template <class P>
ZKT_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub<P>(fe_zero<P>(), a);
}

template <class P>
ZKT_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add<P>(a, a);
}

// arkworks-form Montgomery product a*b*R^-1 mod p on packed operands; defined in fx.hpp.
template <class P>
__host__ __device__ Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b);

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs.  With >= 1 spare bit in the modulus
// the running value stays < 2p, i.e. inside N+1 words, so no (N+2)-th word is carried.
// (kept as the plain 32-bit-limb CIOS reference; the product every kernel uses is fe_mul below,
// implemented on 29-bit limbs in fx.hpp)
template <class P>
ZKT_MUL Fe<P> fe_mul_sat(Fe<P> a, Fe<P> b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = (uint32_t)(x >> 32);
        }
        uint32_t tn = t[N] + c;  // cannot overflow (see bound above)
        uint32_t m = t[0] * P::INV;
        uint64_t x = (uint64_t)m * P::mod(0) + t[0];
        c = (uint32_t)(x >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            x = (uint64_t)m * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = (uint32_t)(x >> 32);
        }
        x = (uint64_t)tn + c;
        t[N - 1] = (uint32_t)x;
        t[N] = (uint32_t)(x >> 32);
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    // t < 2p < 2^(32N): t[N] == 0
    return fe_reduce_once<P>(r);
}

template <class P>
ZKT_HD Fe<P> fe_sqr(const Fe<P>& a) {
    return fe_mul<P>(a, a);
}

template <class P>
ZKT_HD Fe<P> fe_from_mont(const Fe<P>& a) {  // a*R^-1: Montgomery -> canonical
    Fe<P> one = fe_zero<P>();
    one.v[0] = 1;
    return fe_mul<P>(a, one);
}
template <class P>
ZKT_HD Fe<P> fe_to_mont(const Fe<P>& a) {
    return fe_mul<P>(a, fe_r2<P>());
}
template <class P>
ZKT_HD Fe<P> fe_from_u32(uint32_t x) {
    Fe<P> a = fe_zero<P>();
    a.v[0] = x;
    return fe_to_mont<P>(a);
}

// a^e, e a 64-bit exponent (twiddle generation, not on the hot path)
template <class P>
ZKT_HD Fe<P> fe_pow_u64(const Fe<P>& a, uint64_t e) {
    Fe<P> r = fe_one<P>();
    Fe<P> base = a;
    while (e) {
        if (e & 1) r = fe_mul<P>(r, base);
        e >>= 1;
        if (e) base = fe_sqr<P>(base);
    }
    return r;
}

// a^(p-2) (Fermat).  Used for the handful of scalar inversions; bulk inversions go through the
// batched Montgomery-trick kernels.
template <class P>
ZKT_HD Fe<P> fe_inv(const Fe<P>& a) {
    Fe<P> r = fe_one<P>();
    bool started = false;
    uint32_t ex[P::N];  // p - 2
    uint32_t borrow = 2;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        uint32_t m = P::mod(i);
        ex[i] = m - borrow;
        borrow = m < borrow ? 1u : 0u;
    }
#pragma unroll
    for (int li = P::N - 1; li >= 0; --li) {
        uint32_t e = ex[li];
        for (int b = 31; b >= 0; --b) {
            if (started) r = fe_sqr<P>(r);
            if ((e >> b) & 1u) {
                r = fe_mul<P>(r, a);
                started = true;
            }
        }
    }
    return r;
}

// 128-bit wide global / LDS moves of a field element
template <class P>
ZKT_D Fe<P> fe_load(const Fe<P>* p) {
    Fe<P> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) {
        uint4 w = q[i];
        r.v[4 * i + 0] = w.x;
        r.v[4 * i + 1] = w.y;
        r.v[4 * i + 2] = w.z;
        r.v[4 * i + 3] = w.w;
    }
    return r;
}
template <class P>
ZKT_D void fe_store(Fe<P>* p, const Fe<P>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; ++i) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace zkt
