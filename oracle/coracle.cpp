// CPU oracle, C++ part (TEST INFRASTRUCTURE ONLY -- never linked into the product library).
//
// Fast restatement of the third-party arithmetic the reference's prover calls
// (crates absent from /root/reference, conventions restated from their published algorithms):
//   * ark-ff 0.3   Montgomery Fp256 / Fp384 (little-endian u64 limbs, R = 2^(64*limbs))
//   * ark-poly 0.3 Radix2EvaluationDomain fft / ifft / coset_fft / coset_ifft
//                  (call sites plonk-core/src/util.rs:71,85,98,112,125,139)
//   * ark-ec 0.3   VariableBaseMSM::multi_scalar_mul (bucket method, window
//                  c = 3 if n < 32 else floor(log2 n)*69/100 + 2, windows in parallel;
//                  call sites plonk-core/src/commitment.rs:42,78 and kzg10 commit/open)
// It is pinned against the Python big-integer oracle (oracle/*.py) in tests/test_coracle.py and
// serves as (a) the fast checker for GPU parity tests and (b) bench.py's cpu_baseline ("port").
//
// Build: make -C oracle   (g++ -O3 -fopenmp -shared)
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

template <int N>
struct FpParams {
    u64 p[N];
    u64 inv;     // -p^-1 mod 2^64
    u64 r[N];    // R mod p      (Montgomery one)
    u64 r2[N];   // R^2 mod p
};

template <int N>
struct Fp {
    u64 v[N];
};

template <int N>
static inline bool geq(const u64* a, const u64* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}

template <int N>
static inline void sub_noborrow(u64* a, const u64* b) {
    u64 borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}

template <int N>
static inline Fp<N> fadd(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    Fp<N> r;
    u64 carry = 0;
    for (int i = 0; i < N; ++i) {
        u128 s = (u128)a.v[i] + b.v[i] + carry;
        r.v[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    if (carry || geq<N>(r.v, P.p)) sub_noborrow<N>(r.v, P.p);
    return r;
}

template <int N>
static inline Fp<N> fsub(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    Fp<N> r = a;
    if (!geq<N>(a.v, b.v)) {
        u64 carry = 0;
        for (int i = 0; i < N; ++i) {
            u128 s = (u128)r.v[i] + P.p[i] + carry;
            r.v[i] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
    sub_noborrow<N>(r.v, b.v);
    return r;
}

template <int N>
static inline Fp<N> fneg(const Fp<N>& a, const FpParams<N>& P) {
    bool z = true;
    for (int i = 0; i < N; ++i) z &= (a.v[i] == 0);
    if (z) return a;
    Fp<N> r;
    memcpy(r.v, P.p, sizeof(r.v));
    sub_noborrow<N>(r.v, a.v);
    return r;
}

// CIOS Montgomery multiplication
template <int N>
static inline Fp<N> fmul(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    u64 t[N + 2] = {0};
    for (int i = 0; i < N; ++i) {
        u64 c = 0;
        for (int j = 0; j < N; ++j) {
            u128 x = (u128)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (u64)x;
            c = (u64)(x >> 64);
        }
        u128 x = (u128)t[N] + c;
        t[N] = (u64)x;
        t[N + 1] = (u64)(x >> 64);
        u64 m = t[0] * P.inv;
        x = (u128)m * P.p[0] + t[0];
        c = (u64)(x >> 64);
        for (int j = 1; j < N; ++j) {
            x = (u128)m * P.p[j] + t[j] + c;
            t[j - 1] = (u64)x;
            c = (u64)(x >> 64);
        }
        x = (u128)t[N] + c;
        t[N - 1] = (u64)x;
        t[N] = t[N + 1] + (u64)(x >> 64);
    }
    Fp<N> r;
    memcpy(r.v, t, sizeof(r.v));
    if (t[N] || geq<N>(r.v, P.p)) sub_noborrow<N>(r.v, P.p);
    return r;
}

template <int N>
static inline bool fis_zero(const Fp<N>& a) {
    u64 o = 0;
    for (int i = 0; i < N; ++i) o |= a.v[i];
    return o == 0;
}
template <int N>
static inline bool feq(const Fp<N>& a, const Fp<N>& b) {
    u64 o = 0;
    for (int i = 0; i < N; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

template <int N>
static Fp<N> fpow(const Fp<N>& a, const u64* e, int elimbs, const FpParams<N>& P) {
    Fp<N> r;
    memcpy(r.v, P.r, sizeof(r.v));
    bool started = false;
    for (int i = elimbs * 64 - 1; i >= 0; --i) {
        if (started) r = fmul<N>(r, r, P);
        if ((e[i / 64] >> (i % 64)) & 1) {
            r = fmul<N>(r, a, P);
            started = true;
        }
    }
    return r;
}

template <int N>
static Fp<N> finv(const Fp<N>& a, const FpParams<N>& P) {  // Fermat: a^(p-2)
    u64 e[N];
    memcpy(e, P.p, sizeof(e));
    u64 borrow = 2;
    for (int i = 0; i < N && borrow; ++i) {
        u64 old = e[i];
        e[i] -= borrow;
        borrow = old < borrow ? 1 : 0;
    }
    return fpow<N>(a, e, N, P);
}

template <int N>
static inline Fp<N> from_u64(u64 x, const FpParams<N>& P) {
    Fp<N> a{};
    a.v[0] = x;
    Fp<N> r2;
    memcpy(r2.v, P.r2, sizeof(r2.v));
    return fmul<N>(a, r2, P);
}
template <int N>
static inline Fp<N> to_mont(const Fp<N>& a, const FpParams<N>& P) {
    Fp<N> r2;
    memcpy(r2.v, P.r2, sizeof(r2.v));
    return fmul<N>(a, r2, P);
}
template <int N>
static inline Fp<N> from_mont(const Fp<N>& a, const FpParams<N>& P) {
    Fp<N> one{};
    one.v[0] = 1;
    return fmul<N>(a, one, P);
}

// ---------------------------------------------------------------------------------------------
// Parameter sets (values restated from ark-bn254 / ark-bls12-381 0.3; checked against Python
// big integers in tests/test_coracle.py through orc_params()).
// ---------------------------------------------------------------------------------------------
static const FpParams<4> BN254_FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
static const FpParams<4> BN254_FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
static const FpParams<4> BLS_FR = {
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
    0xfffffffeffffffffULL,
    {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL},
    {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const FpParams<6> BLS_FQ = {
    {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
     0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL},
    0x89f3fffcfffcfffdULL,
    {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
     0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL},
    {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
     0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};

struct FrInfo {
    const FpParams<4>* P;
    int two_adicity;
    u64 generator;
};
static const FrInfo FR_INFO[2] = {{&BN254_FR, 28, 5}, {&BLS_FR, 32, 7}};

// ---------------------------------------------------------------------------------------------
// NTT (ark-poly 0.3 Radix2EvaluationDomain)
// ---------------------------------------------------------------------------------------------
typedef Fp<4> Fr;

static Fr fr_root_of_unity(const FrInfo& I, int log_n) {
    const FpParams<4>& P = *I.P;
    // TWO_ADIC_ROOT_OF_UNITY = g^((p-1)/2^s)
    u64 e[4];
    memcpy(e, P.p, sizeof(e));
    e[0] -= 1;
    int s = I.two_adicity;
    // shift right by s
    for (int k = 0; k < s; ++k) {
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? (e[i + 1] << 63) : 0);
    }
    Fr w = fpow<4>(from_u64<4>(I.generator, P), e, 4, P);
    for (int k = 0; k < s - log_n; ++k) w = fmul<4>(w, w, P);
    return w;
}

static inline size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; ++i) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

static void ntt_in_place(Fr* a, int log_n, const Fr& w, const FpParams<4>& P) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    // twiddle table w^0..w^(n/2-1)
    std::vector<Fr> tw(n / 2);
    {
        Fr one;
        memcpy(one.v, P.r, sizeof(one.v));
        const size_t CH = 1024;
        size_t nchunks = (n / 2 + CH - 1) / CH;
#pragma omp parallel for schedule(static)
        for (long c = 0; c < (long)nchunks; ++c) {
            size_t lo = c * CH, hi = std::min(n / 2, lo + CH);
            u64 e[1] = {lo};
            Fr cur = fpow<4>(w, e, 1, P);
            for (size_t i = lo; i < hi; ++i) {
                tw[i] = cur;
                cur = fmul<4>(cur, w, P);
            }
        }
    }
    // DIF: natural in, bit-reversed out (ark-poly io_helper), then derange (bit reversal)
    for (int s = 0; s < log_n; ++s) {
        size_t m = n >> s;  // block size
        size_t half = m >> 1;
        size_t step = n / m;  // twiddle stride
#pragma omp parallel for schedule(static)
        for (long idx = 0; idx < (long)(n / 2); ++idx) {
            size_t blk = idx / half, j = idx % half;
            size_t i0 = blk * m + j, i1 = i0 + half;
            Fr u = a[i0], v = a[i1];
            a[i0] = fadd<4>(u, v, P);
            Fr d = fsub<4>(u, v, P);
            a[i1] = j ? fmul<4>(d, tw[j * step], P) : d;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, log_n);
        if (i < j) std::swap(a[i], a[j]);
    }
}

extern "C" int orc_ntt(int curve, int log_n, int inverse, int coset, const u64* in, size_t in_len, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FrInfo& I = FR_INFO[curve];
    const FpParams<4>& P = *I.P;
    if (log_n < 0 || log_n > I.two_adicity) return 2;
    size_t n = (size_t)1 << log_n;
    if (in_len > n) return 3;
    Fr* a = (Fr*)out;
    if ((const void*)in != (const void*)out) memcpy(a, in, in_len * sizeof(Fr));
    memset(a + in_len, 0, (n - in_len) * sizeof(Fr));
    Fr w = fr_root_of_unity(I, log_n);
    Fr g = from_u64<4>(I.generator, P);
    auto distribute = [&](const Fr& base, const Fr* scale) {
        const size_t CH = 1024;
        size_t nchunks = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static)
        for (long c = 0; c < (long)nchunks; ++c) {
            size_t lo = c * CH, hi = std::min(n, lo + CH);
            u64 e[1] = {lo};
            Fr cur = fpow<4>(base, e, 1, P);
            if (scale) cur = fmul<4>(cur, *scale, P);
            for (size_t i = lo; i < hi; ++i) {
                a[i] = fmul<4>(a[i], cur, P);
                cur = fmul<4>(cur, base, P);
            }
        }
    };
    if (!inverse) {
        if (coset) distribute(g, nullptr);
        ntt_in_place(a, log_n, w, P);
    } else {
        Fr winv = finv<4>(w, P);
        ntt_in_place(a, log_n, winv, P);
        Fr ninv = finv<4>(from_u64<4>((u64)n, P), P);
        if (coset) {
            Fr ginv = finv<4>(g, P);
            distribute(ginv, &ninv);
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long)n; ++i) a[i] = fmul<4>(a[i], ninv, P);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// G1 (short Weierstrass, a = 0), Jacobian coordinates as in ark-ec 0.3
// ---------------------------------------------------------------------------------------------
template <int N>
struct Aff {
    Fp<N> x, y;
    bool inf;
};
template <int N>
struct Jac {
    Fp<N> x, y, z;
};

template <int N>
static inline Jac<N> jzero(const FpParams<N>& P) {
    Jac<N> r;
    memcpy(r.x.v, P.r, sizeof(r.x.v));
    memcpy(r.y.v, P.r, sizeof(r.y.v));
    memset(r.z.v, 0, sizeof(r.z.v));
    return r;
}

template <int N>
static inline void jdouble(Jac<N>& p, const FpParams<N>& P) {  // dbl-2009-l
    if (fis_zero<N>(p.z)) return;
    Fp<N> a = fmul<N>(p.x, p.x, P);
    Fp<N> b = fmul<N>(p.y, p.y, P);
    Fp<N> c = fmul<N>(b, b, P);
    Fp<N> t = fadd<N>(p.x, b, P);
    t = fmul<N>(t, t, P);
    t = fsub<N>(fsub<N>(t, a, P), c, P);
    Fp<N> d = fadd<N>(t, t, P);
    Fp<N> e = fadd<N>(fadd<N>(a, a, P), a, P);
    Fp<N> f = fmul<N>(e, e, P);
    Fp<N> z3 = fmul<N>(p.z, p.y, P);
    z3 = fadd<N>(z3, z3, P);
    Fp<N> x3 = fsub<N>(fsub<N>(f, d, P), d, P);
    Fp<N> c8 = fadd<N>(c, c, P);
    c8 = fadd<N>(c8, c8, P);
    c8 = fadd<N>(c8, c8, P);
    Fp<N> y3 = fsub<N>(fmul<N>(fsub<N>(d, x3, P), e, P), c8, P);
    p.x = x3;
    p.y = y3;
    p.z = z3;
}

template <int N>
static inline void jadd_mixed(Jac<N>& p, const Aff<N>& q, const FpParams<N>& P) {  // madd-2007-bl
    if (q.inf) return;
    if (fis_zero<N>(p.z)) {
        p.x = q.x;
        p.y = q.y;
        memcpy(p.z.v, P.r, sizeof(p.z.v));
        return;
    }
    Fp<N> z1z1 = fmul<N>(p.z, p.z, P);
    Fp<N> u2 = fmul<N>(q.x, z1z1, P);
    Fp<N> s2 = fmul<N>(fmul<N>(q.y, p.z, P), z1z1, P);
    if (feq<N>(p.x, u2)) {
        if (feq<N>(p.y, s2)) {
            jdouble<N>(p, P);
        } else {
            p = jzero<N>(P);
        }
        return;
    }
    Fp<N> h = fsub<N>(u2, p.x, P);
    Fp<N> hh = fmul<N>(h, h, P);
    Fp<N> i = fadd<N>(hh, hh, P);
    i = fadd<N>(i, i, P);
    Fp<N> j = fmul<N>(h, i, P);
    Fp<N> r = fsub<N>(s2, p.y, P);
    r = fadd<N>(r, r, P);
    Fp<N> v = fmul<N>(p.x, i, P);
    Fp<N> x3 = fsub<N>(fsub<N>(fsub<N>(fmul<N>(r, r, P), j, P), v, P), v, P);
    Fp<N> yj = fmul<N>(p.y, j, P);
    yj = fadd<N>(yj, yj, P);
    Fp<N> y3 = fsub<N>(fmul<N>(r, fsub<N>(v, x3, P), P), yj, P);
    Fp<N> zh = fadd<N>(p.z, h, P);
    Fp<N> z3 = fsub<N>(fsub<N>(fmul<N>(zh, zh, P), z1z1, P), hh, P);
    p.x = x3;
    p.y = y3;
    p.z = z3;
}

template <int N>
static inline void jadd(Jac<N>& p, const Jac<N>& q, const FpParams<N>& P) {  // add-2007-bl
    if (fis_zero<N>(q.z)) return;
    if (fis_zero<N>(p.z)) {
        p = q;
        return;
    }
    Fp<N> z1z1 = fmul<N>(p.z, p.z, P);
    Fp<N> z2z2 = fmul<N>(q.z, q.z, P);
    Fp<N> u1 = fmul<N>(p.x, z2z2, P);
    Fp<N> u2 = fmul<N>(q.x, z1z1, P);
    Fp<N> s1 = fmul<N>(fmul<N>(p.y, q.z, P), z2z2, P);
    Fp<N> s2 = fmul<N>(fmul<N>(q.y, p.z, P), z1z1, P);
    if (feq<N>(u1, u2)) {
        if (feq<N>(s1, s2)) {
            jdouble<N>(p, P);
        } else {
            p = jzero<N>(P);
        }
        return;
    }
    Fp<N> h = fsub<N>(u2, u1, P);
    Fp<N> i = fadd<N>(h, h, P);
    i = fmul<N>(i, i, P);
    Fp<N> j = fmul<N>(h, i, P);
    Fp<N> r = fsub<N>(s2, s1, P);
    r = fadd<N>(r, r, P);
    Fp<N> v = fmul<N>(u1, i, P);
    Fp<N> x3 = fsub<N>(fsub<N>(fsub<N>(fmul<N>(r, r, P), j, P), v, P), v, P);
    Fp<N> sj = fmul<N>(s1, j, P);
    sj = fadd<N>(sj, sj, P);
    Fp<N> y3 = fsub<N>(fmul<N>(r, fsub<N>(v, x3, P), P), sj, P);
    Fp<N> zz = fadd<N>(p.z, q.z, P);
    Fp<N> z3 = fmul<N>(fsub<N>(fsub<N>(fmul<N>(zz, zz, P), z1z1, P), z2z2, P), h, P);
    p.x = x3;
    p.y = y3;
    p.z = z3;
}

template <int N>
static Aff<N> jto_affine(const Jac<N>& p, const FpParams<N>& P) {
    Aff<N> r;
    if (fis_zero<N>(p.z)) {
        memset(&r, 0, sizeof(r));
        memcpy(r.y.v, P.r, sizeof(r.y.v));  // GroupAffine::zero() = (0, 1, inf)
        r.inf = true;
        return r;
    }
    Fp<N> zi = finv<N>(p.z, P);
    Fp<N> zi2 = fmul<N>(zi, zi, P);
    r.x = fmul<N>(p.x, zi2, P);
    r.y = fmul<N>(fmul<N>(p.y, zi2, P), zi, P);
    r.inf = false;
    return r;
}

static inline int ln_without_floats(size_t a) {
    int lg = 0;
    while ((a >> (lg + 1)) != 0) ++lg;
    return lg * 69 / 100;
}

static inline u64 scalar_window(const u64* s, int start, int c) {
    // (scalar >> start) mod 2^c, c <= 32
    int limb = start / 64, off = start % 64;
    u64 v = s[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
    return v & (((u64)1 << c) - 1);
}

// VariableBaseMSM::multi_scalar_mul restated.  scalars are canonical (non-Montgomery) bigints.
template <int N>
static Jac<N> msm_pippenger(const Aff<N>* bases, const u64* scalars, size_t size, int num_bits,
                            const FpParams<N>& P) {
    int c = size < 32 ? 3 : ln_without_floats(size) + 2;
    int nwin = (num_bits + c - 1) / c;
    std::vector<Jac<N>> window_sums(nwin);
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < nwin; ++w) {
        int w_start = w * c;
        Jac<N> res = jzero<N>(P);
        std::vector<Jac<N>> buckets(((size_t)1 << c) - 1, jzero<N>(P));
        for (size_t i = 0; i < size; ++i) {
            const u64* s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
            if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
                if (w_start == 0) jadd_mixed<N>(res, bases[i], P);
            } else {
                u64 d = scalar_window(s, w_start, c);
                if (d != 0) jadd_mixed<N>(buckets[d - 1], bases[i], P);
            }
        }
        Jac<N> running = jzero<N>(P);
        for (size_t b = buckets.size(); b-- > 0;) {
            jadd<N>(running, buckets[b], P);
            jadd<N>(res, running, P);
        }
        window_sums[w] = res;
    }
    Jac<N> total = jzero<N>(P);
    for (int w = nwin - 1; w >= 1; --w) {
        jadd<N>(total, window_sums[w], P);
        for (int k = 0; k < c; ++k) jdouble<N>(total, P);
    }
    jadd<N>(total, window_sums[0], P);
    return total;
}

template <int N>
static int msm_entry(const FpParams<N>& PQ, const FrInfo& I, int fr_bits, const u64* bases_xy, const u64* scalars,
                     size_t n, int scalars_mont, u64* out_xy, int* out_inf) {
    std::vector<Aff<N>> bases(n);
    for (size_t i = 0; i < n; ++i) {
        memcpy(bases[i].x.v, bases_xy + 2 * N * i, N * 8);
        memcpy(bases[i].y.v, bases_xy + 2 * N * i + N, N * 8);
        bases[i].inf = fis_zero<N>(bases[i].x) && fis_zero<N>(bases[i].y);  // (0,0) encodes infinity on this ABI
    }
    std::vector<u64> sc(4 * n);
    if (scalars_mont) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; ++i) {
            Fr s;
            memcpy(s.v, scalars + 4 * i, 32);
            s = from_mont<4>(s, *I.P);
            memcpy(&sc[4 * i], s.v, 32);
        }
    } else {
        memcpy(sc.data(), scalars, 32 * n);
    }
    Jac<N> r = msm_pippenger<N>(bases.data(), sc.data(), n, fr_bits, PQ);
    Aff<N> a = jto_affine<N>(r, PQ);
    memcpy(out_xy, a.x.v, N * 8);
    memcpy(out_xy + N, a.y.v, N * 8);
    *out_inf = a.inf ? 1 : 0;
    return 0;
}

// bases: n affine points, x||y Montgomery limbs (4+4 or 6+6 u64), (0,0) = infinity.
extern "C" int orc_msm(int curve, const u64* bases_xy, const u64* scalars, size_t n, int scalars_mont, u64* out_xy,
                       int* out_inf) {
    if (curve == 0) return msm_entry<4>(BN254_FQ, FR_INFO[0], 254, bases_xy, scalars, n, scalars_mont, out_xy, out_inf);
    if (curve == 1) return msm_entry<6>(BLS_FQ, FR_INFO[1], 255, bases_xy, scalars, n, scalars_mont, out_xy, out_inf);
    return 1;
}

// powers_of_g of a KZG SRS with a known test trapdoor: out[i] = tau^i * G (affine, Montgomery).
template <int N>
static int srs_entry(const FpParams<N>& PQ, const FrInfo& I, const u64* gx, const u64* gy, const u64* tau,
                     size_t count, u64* out_xy) {
    const FpParams<4>& PR = *I.P;
    Aff<N> g;
    memcpy(g.x.v, gx, N * 8);
    memcpy(g.y.v, gy, N * 8);
    g.x = to_mont<N>(g.x, PQ);
    g.y = to_mont<N>(g.y, PQ);
    g.inf = false;
    // fixed-base table: T[w][d] = d * 2^(8w) * G, d in 1..255, kept Jacobian then normalised
    const int W = 32;
    std::vector<Aff<N>> table((size_t)W * 256);
    {
        Jac<N> base = jzero<N>(PQ);
        jadd_mixed<N>(base, g, PQ);
        for (int w = 0; w < W; ++w) {
            Aff<N> b = jto_affine<N>(base, PQ);
            Jac<N> acc = jzero<N>(PQ);
            table[(size_t)w * 256].inf = true;
            for (int d = 1; d < 256; ++d) {
                jadd_mixed<N>(acc, b, PQ);
                table[(size_t)w * 256 + d] = jto_affine<N>(acc, PQ);
            }
            for (int k = 0; k < 8; ++k) jdouble<N>(base, PQ);
        }
    }
    Fr t;
    memcpy(t.v, tau, 32);
    t = to_mont<4>(t, PR);
    const size_t CH = 256;
    size_t nchunks = (count + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
    for (long c = 0; c < (long)nchunks; ++c) {
        size_t lo = c * CH, hi = std::min(count, lo + CH);
        u64 e[1] = {lo};
        Fr cur = fpow<4>(t, e, 1, PR);
        for (size_t i = lo; i < hi; ++i) {
            Fr s = from_mont<4>(cur, PR);
            Jac<N> acc = jzero<N>(PQ);
            for (int w = 0; w < W; ++w) {
                unsigned d = (unsigned)((s.v[w / 8] >> (8 * (w % 8))) & 0xff);
                if (d) jadd_mixed<N>(acc, table[(size_t)w * 256 + d], PQ);
            }
            Aff<N> a = jto_affine<N>(acc, PQ);
            if (a.inf) {
                memset(out_xy + 2 * N * i, 0, 2 * N * 8);
            } else {
                memcpy(out_xy + 2 * N * i, a.x.v, N * 8);
                memcpy(out_xy + 2 * N * i + N, a.y.v, N * 8);
            }
            cur = fmul<4>(cur, t, PR);
        }
    }
    return 0;
}

extern "C" int orc_srs(int curve, const u64* gx, const u64* gy, const u64* tau, size_t count, u64* out_xy) {
    if (curve == 0) return srs_entry<4>(BN254_FQ, FR_INFO[0], gx, gy, tau, count, out_xy);
    if (curve == 1) return srs_entry<6>(BLS_FQ, FR_INFO[1], gx, gy, tau, count, out_xy);
    return 1;
}

// elementwise helpers on Fr arrays (Montgomery in memory)
extern "C" int orc_fr_convert(int curve, int to_montgomery, const u64* in, size_t n, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        Fr a;
        memcpy(a.v, in + 4 * i, 32);
        a = to_montgomery ? to_mont<4>(a, P) : from_mont<4>(a, P);
        memcpy(out + 4 * i, a.v, 32);
    }
    return 0;
}

extern "C" int orc_fq_convert(int curve, int to_montgomery, const u64* in, size_t n, u64* out) {
    if (curve == 0) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; ++i) {
            Fp<4> a;
            memcpy(a.v, in + 4 * i, 32);
            a = to_montgomery ? to_mont<4>(a, BN254_FQ) : from_mont<4>(a, BN254_FQ);
            memcpy(out + 4 * i, a.v, 32);
        }
        return 0;
    }
    if (curve == 1) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; ++i) {
            Fp<6> a;
            memcpy(a.v, in + 6 * i, 48);
            a = to_montgomery ? to_mont<6>(a, BLS_FQ) : from_mont<6>(a, BLS_FQ);
            memcpy(out + 6 * i, a.v, 48);
        }
        return 0;
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------
// The O(n) loops of the prover (plonk-core/src/proof_system/prove.rs:59-470), restated on Montgomery
// arrays so that whole proofs at n = 2^14 .. 2^22 are available as the byte-exact check of the GPU
// path.  Each function is pinned against its big-integer twin in oracle/plonk.py at n <= 4096
// (tests/test_coracle.py).  Field arithmetic is exact, so evaluation order (batched inversions,
// OpenMP chunking) cannot change a result.
// ---------------------------------------------------------------------------------------------
static inline Fr fr_load(const u64* p) {
    Fr r;
    memcpy(r.v, p, 32);
    return r;
}
static inline void fr_store(u64* p, const Fr& a) { memcpy(p, a.v, 32); }
static inline Fr fr_one(const FpParams<4>& P) {
    Fr r;
    memcpy(r.v, P.r, 32);
    return r;
}
static inline Fr fr_pow_u64(const Fr& a, u64 e, const FpParams<4>& P) {
    u64 ee[1] = {e};
    return fpow<4>(a, ee, 1, P);
}

// Montgomery's trick per chunk; returns false when some element is zero (the reference unwraps an inverse there)
static bool batch_inverse(Fr* v, size_t n, const FpParams<4>& P) {
    const size_t CH = 4096;
    size_t nchunks = (n + CH - 1) / CH;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (long c = 0; c < (long)nchunks; ++c) {
        size_t lo = c * CH, hi = std::min(n, lo + CH);
        std::vector<Fr> pre(hi - lo);
        Fr acc = fr_one(P);
        for (size_t i = lo; i < hi; ++i) {
            if (fis_zero<4>(v[i])) bad = 1;
            pre[i - lo] = acc;
            acc = fmul<4>(acc, v[i], P);
        }
        if (bad) continue;
        Fr inv = finv<4>(acc, P);
        for (size_t i = hi; i-- > lo;) {
            Fr t = fmul<4>(inv, pre[i - lo], P);
            inv = fmul<4>(inv, v[i], P);
            v[i] = t;
        }
    }
    return !bad;
}

// out[i] = w^i * scale, i < n (domain elements / coset points)
static void fill_powers(Fr* out, size_t n, const Fr& w, const Fr& scale, const FpParams<4>& P) {
    const size_t CH = 4096;
    size_t nchunks = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)nchunks; ++c) {
        size_t lo = c * CH, hi = std::min(n, lo + CH);
        Fr cur = fmul<4>(fr_pow_u64(w, lo, P), scale, P);
        for (size_t i = lo; i < hi; ++i) {
            out[i] = cur;
            cur = fmul<4>(cur, w, P);
        }
    }
}

// kind 0: domain.elements() (util.rs:27-50); kind 1: coset points g * w^i (keys/mod.rs:110-113 x_coset)
extern "C" int orc_domain_points(int curve, int log_n, int kind, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FrInfo& I = FR_INFO[curve];
    const FpParams<4>& P = *I.P;
    if (log_n < 0 || log_n > I.two_adicity) return 2;
    Fr w = fr_root_of_unity(I, log_n);
    Fr scale = kind ? from_u64<4>(I.generator, P) : fr_one(P);
    fill_powers((Fr*)out, (size_t)1 << log_n, w, scale, P);
    return 0;
}

// elementwise: op 0 add, 1 sub, 2 mul (prove.rs:157-161 f = q_lookup . c uses the product)
extern "C" int orc_fr_vec_op(int curve, int op, const u64* a, const u64* b, size_t n, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        Fr x = fr_load(a + 4 * i), y = fr_load(b + 4 * i);
        Fr r = op == 0 ? fadd<4>(x, y, P) : op == 1 ? fsub<4>(x, y, P) : fmul<4>(x, y, P);
        fr_store(out + 4 * i, r);
    }
    return 0;
}

// permutation/mod.rs:181-254 compute_permutation_poly, evaluation form (before the iFFT at :256).
// z[0] = 1, z[i+1] = z[i] * num_i / den_i for i < n - 1.  Returns 6 when a denominator is zero.
extern "C" int orc_z1_evals(int curve, int log_n, const u64* beta_, const u64* gamma_, const u64* a, const u64* b,
                            const u64* c, const u64* s1, const u64* s2, const u64* s3, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FrInfo& I = FR_INFO[curve];
    const FpParams<4>& P = *I.P;
    size_t n = (size_t)1 << log_n;
    Fr beta = fr_load(beta_), gamma = fr_load(gamma_);
    Fr k1 = from_u64<4>(7, P), k2 = from_u64<4>(13, P);   // permutation/constants.rs:13-20
    std::vector<Fr> roots(n), num(n), den(n);
    fill_powers(roots.data(), n, fr_root_of_unity(I, log_n), fr_one(P), P);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)(n - 1); ++i) {
        Fr br = fmul<4>(beta, roots[i], P);
        Fr ai = fr_load(a + 4 * i), bi = fr_load(b + 4 * i), ci = fr_load(c + 4 * i);
        Fr ag = fadd<4>(ai, gamma, P), bg = fadd<4>(bi, gamma, P), cg = fadd<4>(ci, gamma, P);
        Fr nu = fmul<4>(fadd<4>(br, ag, P), fadd<4>(fmul<4>(k1, br, P), bg, P), P);
        num[i] = fmul<4>(nu, fadd<4>(fmul<4>(k2, br, P), cg, P), P);
        Fr de = fmul<4>(fadd<4>(fmul<4>(beta, fr_load(s1 + 4 * i), P), ag, P),
                        fadd<4>(fmul<4>(beta, fr_load(s2 + 4 * i), P), bg, P), P);
        den[i] = fmul<4>(de, fadd<4>(fmul<4>(beta, fr_load(s3 + 4 * i), P), cg, P), P);
    }
    if (n > 1 && !batch_inverse(den.data(), n - 1, P)) return 6;
    Fr state = fr_one(P);
    fr_store(out, state);
    for (size_t i = 0; i + 1 < n; ++i) {
        state = fmul<4>(state, fmul<4>(num[i], den[i], P), P);
        fr_store(out + 4 * (i + 1), state);
    }
    return 0;
}

// lookup/mod.rs:94-151 compute_lookup_permutation_poly, evaluation form (before the iFFT at :153)
extern "C" int orc_z2_evals(int curve, int log_n, const u64* delta_, const u64* epsilon_, const u64* f, const u64* t,
                            const u64* h1, const u64* h2, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
    size_t n = (size_t)1 << log_n;
    Fr delta = fr_load(delta_), eps = fr_load(epsilon_);
    Fr opd = fadd<4>(fr_one(P), delta, P), eopd = fmul<4>(eps, opd, P);
    std::vector<Fr> num(n), den(n);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)(n - 1); ++i) {
        Fr ti = fr_load(t + 4 * i), tn = fr_load(t + 4 * (i + 1));
        Fr h1i = fr_load(h1 + 4 * i), h1n = fr_load(h1 + 4 * (i + 1)), h2i = fr_load(h2 + 4 * i);
        Fr nu = fmul<4>(opd, fadd<4>(eps, fr_load(f + 4 * i), P), P);
        num[i] = fmul<4>(nu, fadd<4>(fadd<4>(fmul<4>(delta, tn, P), eopd, P), ti, P), P);
        den[i] = fmul<4>(fadd<4>(fadd<4>(fmul<4>(delta, h2i, P), eopd, P), h1i, P),
                         fadd<4>(fadd<4>(fmul<4>(delta, h1n, P), eopd, P), h2i, P), P);
    }
    if (n > 1 && !batch_inverse(den.data(), n - 1, P)) return 6;
    Fr state = fr_one(P);
    fr_store(out, state);
    for (size_t i = 0; i + 1 < n; ++i) {
        state = fmul<4>(state, fmul<4>(num[i], den[i], P), P);
        fr_store(out + 4 * (i + 1), state);
    }
    return 0;
}

// proof_system/quotient_poly.rs:98-224 with keys/arithmetic.rs:67-81, keys/permutation.rs:97-137,
// keys/lookup.rs:81-122.  epk: q_m q_l q_r q_o q_c q_lookup q_table sigma1 sigma2 sigma3 x zh l_1;
// wit: a b c pi z1 z2 t h1 h2; all vectors are 4n coset evaluations; "next" = index + 4 (quotient_poly.rs:52-94).
extern "C" int orc_quotient_evals(int curve, int log_n, const u64* ch, const u64* const* epk, const u64* const* wit,
                                  u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
    size_t N = (size_t)4 << log_n;
    Fr alpha = fr_load(ch), beta = fr_load(ch + 4), gamma = fr_load(ch + 8), delta = fr_load(ch + 12),
       eps = fr_load(ch + 16);
    Fr one = fr_one(P);
    Fr a2 = fmul<4>(alpha, alpha, P), a3 = fmul<4>(a2, alpha, P), a4 = fmul<4>(a3, alpha, P), a5 = fmul<4>(a4, alpha, P);
    Fr opd = fadd<4>(delta, one, P), eopd = fmul<4>(eps, opd, P);
    Fr k1 = from_u64<4>(7, P), k2 = from_u64<4>(13, P);
    Fr nalpha = fneg<4>(alpha, P), na3 = fneg<4>(a3, P);
    std::vector<Fr> zhinv(N);
    memcpy(zhinv.data(), epk[11], N * 32);
    if (!batch_inverse(zhinv.data(), N, P)) return 6;
    enum { QM, QL, QR, QO, QC, QLK, QT, S1, S2, S3, X, ZH, L1 };
    enum { A, B, C, PI, Z1, Z2, T, H1, H2 };
#define E(k) fr_load(epk[k] + 4 * i)
#define W(k) fr_load(wit[k] + 4 * i)
#define WN(k) fr_load(wit[k] + 4 * j)
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
        size_t j = ((size_t)i + 4) % N;
        Fr a = W(A), b = W(B), c = W(C);
        Fr arith = fmul<4>(fmul<4>(a, b, P), E(QM), P);
        arith = fadd<4>(arith, fmul<4>(a, E(QL), P), P);
        arith = fadd<4>(arith, fmul<4>(b, E(QR), P), P);
        arith = fadd<4>(arith, fmul<4>(c, E(QO), P), P);
        arith = fadd<4>(fadd<4>(arith, E(QC), P), W(PI), P);
        Fr ag = fadd<4>(a, gamma, P), bg = fadd<4>(b, gamma, P), cg = fadd<4>(c, gamma, P);
        Fr bx = fmul<4>(beta, E(X), P);
        Fr z1 = W(Z1), z1n = WN(Z1), z2 = W(Z2), z2n = WN(Z2);
        Fr p1 = fmul<4>(fmul<4>(alpha, z1, P), fadd<4>(bx, ag, P), P);
        p1 = fmul<4>(fmul<4>(p1, fadd<4>(fmul<4>(bx, k1, P), bg, P), P), fadd<4>(fmul<4>(bx, k2, P), cg, P), P);
        Fr p2 = fmul<4>(fmul<4>(nalpha, z1n, P), fadd<4>(fmul<4>(beta, E(S1), P), ag, P), P);
        p2 = fmul<4>(fmul<4>(p2, fadd<4>(fmul<4>(beta, E(S2), P), bg, P), P), fadd<4>(fmul<4>(beta, E(S3), P), cg, P), P);
        Fr l1 = E(L1);
        Fr p3 = fmul<4>(fmul<4>(fsub<4>(z1, one, P), l1, P), a2, P);
        Fr t = W(T), tn = WN(T), h1 = W(H1), h1n = WN(H1), h2 = W(H2);
        Fr k1_ = fmul<4>(fmul<4>(fmul<4>(a3, z2, P), opd, P), fadd<4>(eps, fmul<4>(E(QLK), c, P), P), P);
        k1_ = fmul<4>(k1_, fadd<4>(fadd<4>(eopd, t, P), fmul<4>(delta, tn, P), P), P);
        Fr k2_ = fmul<4>(fmul<4>(na3, z2n, P), fadd<4>(fadd<4>(eopd, h1, P), fmul<4>(delta, h2, P), P), P);
        k2_ = fmul<4>(k2_, fadd<4>(fadd<4>(eopd, h2, P), fmul<4>(delta, h1n, P), P), P);
        Fr k3_ = fmul<4>(fmul<4>(a4, fsub<4>(z2, one, P), P), l1, P);
        Fr k4_ = fmul<4>(fmul<4>(a5, E(QT), P), t, P);
        Fr tot = fadd<4>(fadd<4>(fadd<4>(arith, p1, P), fadd<4>(p2, p3, P), P),
                         fadd<4>(fadd<4>(k1_, k2_, P), fadd<4>(k3_, k4_, P), P), P);
        fr_store(out + 4 * i, fmul<4>(tot, zhinv[i], P));   // quotient_poly.rs:220-224
    }
#undef E
#undef W
#undef WN
    return 0;
}

// lookup/multiset.rs:103-146 combine_split: counters in first-insertion order of t (IndexMap), every element of
// f must already be a key (Error::ElementNotIndexedInTable -> 8); halves alternate for odd counts.
struct Key32 {
    u64 v[4];
    bool operator==(const Key32& o) const { return !memcmp(v, o.v, 32); }
};
struct Key32Hash {
    size_t operator()(const Key32& k) const {
        u64 h = k.v[0] * 0x9E3779B97F4A7C15ULL;
        h ^= (k.v[1] + 0x7F4A7C15ULL + (h << 6) + (h >> 2));
        h ^= (k.v[2] * 0xBF58476D1CE4E5B9ULL) ^ (k.v[3] * 0x94D049BB133111EBULL);
        return (size_t)h;
    }
};
#include <unordered_map>
extern "C" int orc_combine_split(const u64* t, size_t nt, const u64* f, size_t nf, u64* h1, u64* h2, size_t* lens) {
    std::unordered_map<Key32, size_t, Key32Hash> idx;
    idx.reserve(1024);
    std::vector<Key32> keys;
    std::vector<size_t> counts;
    for (size_t i = 0; i < nt; ++i) {
        Key32 k;
        memcpy(k.v, t + 4 * i, 32);
        auto it = idx.find(k);
        if (it == idx.end()) {
            idx.emplace(k, keys.size());
            keys.push_back(k);
            counts.push_back(1);
        } else {
            ++counts[it->second];
        }
    }
    for (size_t i = 0; i < nf; ++i) {
        Key32 k;
        memcpy(k.v, f + 4 * i, 32);
        auto it = idx.find(k);
        if (it == idx.end()) return 8;
        ++counts[it->second];
    }
    size_t e = 0, o = 0;
    bool parity = false;
    for (size_t k = 0; k < keys.size(); ++k) {
        size_t half = counts[k] / 2;
        for (size_t r = 0; r < half; ++r) {
            memcpy(h1 + 4 * e++, keys[k].v, 32);
            memcpy(h2 + 4 * o++, keys[k].v, 32);
        }
        if (counts[k] & 1) {
            if (parity) {
                memcpy(h2 + 4 * o++, keys[k].v, 32);
                parity = false;
            } else {
                memcpy(h1 + 4 * e++, keys[k].v, 32);
                parity = true;
            }
        }
    }
    lens[0] = e;
    lens[1] = o;
    return 0;
}

// out[i] = sum_k scalars[k] * polys[k][i] for i < out_len (coefficients beyond lens[k] are zero): the scaled sums of
// linearization_poly.rs:19-121 and of the opening combinations (prove.rs:381-451)
extern "C" int orc_lincomb(int curve, int k, const u64* const* polys, const size_t* lens, const u64* scalars,
                           size_t out_len, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
    std::vector<Fr> s(k);
    for (int j = 0; j < k; ++j) s[j] = fr_load(scalars + 4 * j);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)out_len; ++i) {
        Fr acc{};
        for (int j = 0; j < k; ++j)
            if ((size_t)i < lens[j]) acc = fadd<4>(acc, fmul<4>(s[j], fr_load(polys[j] + 4 * i), P), P);
        fr_store(out + 4 * i, acc);
    }
    return 0;
}

// DensePolynomial::evaluate (Horner per chunk, chunks combined with powers of the point)
extern "C" int orc_poly_eval(int curve, const u64* poly, size_t len, const u64* point, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
    Fr x = fr_load(point);
    const size_t CH = 8192;
    size_t nchunks = (len + CH - 1) / CH;
    std::vector<Fr> part(nchunks);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)nchunks; ++c) {
        size_t lo = c * CH, hi = std::min(len, lo + CH);
        Fr acc{};
        for (size_t i = hi; i-- > lo;) acc = fadd<4>(fmul<4>(acc, x, P), fr_load(poly + 4 * i), P);
        part[c] = fmul<4>(acc, fr_pow_u64(x, lo, P), P);
    }
    Fr tot{};
    for (size_t c = 0; c < nchunks; ++c) tot = fadd<4>(tot, part[c], P);
    fr_store(out, tot);
    return 0;
}

// kzg10::open witness polynomial: (p(X) - p(z)) / (X - z), the remainder dropped (call sites prove.rs:381-451);
// out receives len - 1 coefficients
extern "C" int orc_div_linear(int curve, const u64* poly, size_t len, const u64* z_, u64* out) {
    if (curve < 0 || curve > 1) return 1;
    const FpParams<4>& P = *FR_INFO[curve].P;
    Fr z = fr_load(z_);
    Fr carry{};
    for (size_t i = len; i-- > 1;) {
        carry = fadd<4>(fr_load(poly + 4 * i), fmul<4>(z, carry, P), P);
        fr_store(out + 4 * (i - 1), carry);
    }
    return 0;
}

// expose parameter tables so the tests can pin them against Python big integers
extern "C" int orc_params(int which, u64* out /* p, inv, r, r2 flattened */) {
    auto dump4 = [&](const FpParams<4>& P) {
        memcpy(out, P.p, 32);
        out[4] = P.inv;
        memcpy(out + 5, P.r, 32);
        memcpy(out + 9, P.r2, 32);
        return 4;
    };
    auto dump6 = [&](const FpParams<6>& P) {
        memcpy(out, P.p, 48);
        out[6] = P.inv;
        memcpy(out + 7, P.r, 48);
        memcpy(out + 13, P.r2, 48);
        return 6;
    };
    switch (which) {
        case 0: return dump4(BN254_FR);
        case 1: return dump4(BN254_FQ);
        case 2: return dump4(BLS_FR);
        case 3: return dump6(BLS_FQ);
    }
    return 0;
}

extern "C" void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

extern "C" int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
