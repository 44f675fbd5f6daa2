// Polynomial / evaluation-vector kernels of the prover rounds (see poly.hpp for the reference
// lines each one replaces).  All values are Fr in Montgomery form, arkworks layout.
#include "poly.hpp"
#include "hostinv.hpp"

namespace zkt {

template <class P>
ZKT_D Fe<P> arg_fe(const uint32_t* w) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = w[i];
    return r;
}

// ---------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------
template <class P>
__global__ void k_mul_vec(const Fe<P>* a, const Fe<P>* b, Fe<P>* o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store<P>(o + i, fe_mul<P>(fe_load<P>(a + i), fe_load<P>(b + i)));
}

// number of coefficients after stripping trailing zeros (DensePolynomial::from_coefficients_vec)
template <class P>
__global__ void k_trim_len(const Fe<P>* p, size_t lo, size_t n, uint32_t* len, int skip_if_set, Fe<P>* zero_at,
                           int zero_count) {
    // the first launch also clears the few coefficients above the transform's output (they held the previous
    // proof's blinders); nothing in this launch reads them
    if (zero_at && blockIdx.x == 0 && (int)threadIdx.x < zero_count) fe_store<P>(zero_at + threadIdx.x, fe_zero<P>());
    // Scans [lo, n).  The polynomials are dense, so the answer is almost always in the last few hundred
    // coefficients: the caller first scans that tail with one small launch and this launch returns at once
    // when the tail already produced a result (same-address atomics serialise at ~13 ns each).
    if (skip_if_set && *len != 0) return;  // only the follow-up launch may do this: len is final by then
    size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < n && !fe_is_zero<P>(fe_load<P>(p + i))) v = (uint32_t)(i + 1);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_down(v, d);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(len, v);
}

// prove.rs:472-483: coeffs.extend(blinders); coeffs[i] -= blinder[i]
template <class P>
__global__ void k_add_blinders(Fe<P>* p, const uint32_t* len, const Fe<P>* bl, int k) {
    int i = threadIdx.x;
    if (i >= k) return;
    uint32_t L = *len;
    Fe<P> b = fe_load<P>(bl + i);
    fe_store<P>(p + L + i, b);
    __syncthreads();
    // (L + k > k always holds for k <= 3 unless L == 0, where position i was just written)
    Fe<P> cur = fe_load<P>(p + i);
    fe_store<P>(p + i, fe_sub<P>(cur, b));
}

// scalars arrive in the 29-bit-limb R' form (H): A * H = A, two terms per reduction
struct LinCombDev {
    const void* poly[LC_MAX_TERMS];
    uint64_t len[LC_MAX_TERMS];
    uint32_t scalar[LC_MAX_TERMS + 1][9];
    int nterms;
};
template <class P>
__global__ void k_lincomb(LinCombDev a, Fe<P>* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fx<P> acc = fx_zero<P>();
#pragma unroll 1
    for (int k = 0; k < a.nterms; k += 2) {
        Fx<P> p[2], sc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool live = k + h < a.nterms && i < a.len[k + h];
            p[h] = live ? fx_unpack<P>(fe_load<P>((const Fe<P>*)a.poly[live ? k + h : k] + i)) : fx_zero<P>();
#pragma unroll
            for (int w = 0; w < 9; ++w) sc[h].l[w] = a.scalar[k + h][w];
        }
        acc = fx_add<P>(acc, fx_mul2_inl<P>(p[0], sc[0], p[1], sc[1]));   // < 2p each: at most LC_MAX_TERMS p in all
    }
    fe_store<P>(out + i, fx_pack<P>(fx_canon<P>(acc)));
}

// ---------------------------------------------------------------------------------------------
// batched point evaluation: result[k] = poly[k](point[k])
// ---------------------------------------------------------------------------------------------
constexpr int EV_E = 8;
constexpr int EV_SEG = 256 * EV_E;
constexpr int EV_PW = 257;   // x^0 .. x^255 and x^256 per polynomial

// pw[k][t] = point[k]^t for t = 0..256, followed by pw[k][257 + b] = point[k]^(2048 b) for b < nblk
template <class P>
__global__ void k_eval_powers(EvalArgs a, Fe<P>* pw, int nblk) {
    const int k = blockIdx.x;
    const Fe<P> x = arg_fe<P>(a.point[k]);
    Fe<P>* row = pw + (size_t)k * (EV_PW + nblk);
    for (int t = threadIdx.x; t < EV_PW + nblk; t += blockDim.x)
        fe_store<P>(row + t, fe_pow_u64<P>(x, t < EV_PW ? (uint64_t)t : (uint64_t)(t - EV_PW) * EV_SEG));
}

// One workgroup per 2048 consecutive coefficients: thread t takes c[base + 256 j + t], j = 0..7 (coalesced), runs
// Horner in x^256 over j, multiplies by x^t, and the workgroup's sum is multiplied by x^base: one product per
// coefficient.
template <class P>
__global__ __launch_bounds__(256) void k_eval_partial(EvalArgs a, const Fe<P>* pw, Fe<P>* partials, int nblk) {
    __shared__ Fe<P> red[256];
    const int k = blockIdx.y;
    const Fe<P>* poly = (const Fe<P>*)a.poly[k];
    const Fe<P>* row = pw + (size_t)k * (EV_PW + nblk);
    const uint64_t len = a.len[k];
    const uint64_t base = (uint64_t)blockIdx.x * EV_SEG;
    const int t = threadIdx.x;
    Fe<P> acc = fe_zero<P>();
    if (base + t < len) {
        const Fe<P> x256 = fe_load<P>(row + 256);
        Fe<P> c[EV_E];
#pragma unroll
        for (int j = 0; j < EV_E; ++j) {
            const uint64_t i = base + (uint64_t)j * 256 + t;
            c[j] = (i < len) ? fe_load<P>(poly + i) : fe_zero<P>();
        }
#pragma unroll 1
        for (int j = EV_E - 1; j >= 0; --j) acc = fe_add<P>(fe_mul<P>(acc, x256), c[j]);
        acc = fe_mul<P>(acc, fe_load<P>(row + t));
    }
    red[t] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) red[t] = fe_add<P>(red[t], red[t + d]);
        __syncthreads();
    }
    if (t == 0)
        fe_store<P>(partials + (size_t)k * nblk + blockIdx.x,
                    base < len ? fe_mul<P>(red[0], fe_load<P>(row + EV_PW + blockIdx.x)) : fe_zero<P>());
}

template <class P>
__global__ __launch_bounds__(256) void k_eval_final(const Fe<P>* partials, int nblk, Fe<P>* results) {
    __shared__ Fe<P> red[256];
    const int k = blockIdx.x, t = threadIdx.x;
    Fe<P> acc = fe_zero<P>();
    for (int j = t; j < nblk; j += 256) acc = fe_add<P>(acc, fe_load<P>(partials + (size_t)k * nblk + j));
    red[t] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) red[t] = fe_add<P>(red[t], red[t + d]);
        __syncthreads();
    }
    if (t == 0) fe_store<P>(results + k, red[0]);
}

// ---------------------------------------------------------------------------------------------
// scans (prefix product / prefix sum), 1024 elements per block
// ---------------------------------------------------------------------------------------------
struct OpMul {
    template <class P> static ZKT_D Fe<P> apply(const Fe<P>& a, const Fe<P>& b) { return fe_mul<P>(a, b); }
    template <class P> static ZKT_D Fe<P> identity() { return fe_one<P>(); }
};
struct OpAdd {
    template <class P> static ZKT_D Fe<P> apply(const Fe<P>& a, const Fe<P>& b) { return fe_add<P>(a, b); }
    template <class P> static ZKT_D Fe<P> identity() { return fe_zero<P>(); }
};
constexpr int SC_E = 4;
constexpr int SC_BLK = 256 * SC_E;

template <class P, class Op>
__global__ __launch_bounds__(256) void k_scan_local(const Fe<P>* in, Fe<P>* out, size_t n, int reverse, Fe<P>* totals) {
    __shared__ Fe<P> s[256];
    const int t = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * SC_BLK + (size_t)t * SC_E;
    Fe<P> x[SC_E];
#pragma unroll
    for (int e = 0; e < SC_E; ++e) {
        size_t li = i0 + e;
        x[e] = (li < n) ? fe_load<P>(in + (reverse ? n - 1 - li : li)) : Op::template identity<P>();
        if (e) x[e] = Op::template apply<P>(x[e - 1], x[e]);
    }
    s[t] = x[SC_E - 1];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        Fe<P> v = s[t];
        if (t >= d) v = Op::template apply<P>(s[t - d], v);
        __syncthreads();
        s[t] = v;
        __syncthreads();
    }
    Fe<P> pre = (t > 0) ? s[t - 1] : Op::template identity<P>();
#pragma unroll
    for (int e = 0; e < SC_E; ++e) {
        size_t li = i0 + e;
        if (li < n) fe_store<P>(out + (reverse ? n - 1 - li : li), t > 0 ? Op::template apply<P>(pre, x[e]) : x[e]);
    }
    if (t == 255) fe_store<P>(totals + blockIdx.x, s[255]);
}

template <class P, class Op>
__global__ __launch_bounds__(256) void k_scan_apply(Fe<P>* out, size_t n, int reverse, const Fe<P>* block_prefix) {
    const size_t blk = blockIdx.x + 1;  // block 0 needs no fix-up
    const Fe<P> pre = fe_load<P>(block_prefix + blk - 1);
#pragma unroll
    for (int e = 0; e < SC_E; ++e) {
        size_t li = blk * SC_BLK + (size_t)e * 256 + threadIdx.x;
        if (li < n) {
            size_t pi = reverse ? n - 1 - li : li;
            fe_store<P>(out + pi, Op::template apply<P>(pre, fe_load<P>(out + pi)));
        }
    }
}

template <class P, class Op>
static int scan_t(zkt_ctx* c, const Fe<P>* in, Fe<P>* out, size_t n, bool reverse, Fe<P>* tmp) {
    if (n == 0) return ZKT_OK;
    const size_t nb = (n + SC_BLK - 1) / SC_BLK;
    Fe<P>* totals = tmp;
    Fe<P>* prefix = tmp + nb;
    hipLaunchKernelGGL((k_scan_local<P, Op>), dim3((unsigned)nb), dim3(256), 0, c->stream, in, out, n, reverse ? 1 : 0, totals);
    ZKT_HIP(c, hipGetLastError());
    if (nb > 1) {
        int rc = scan_t<P, Op>(c, totals, prefix, nb, false, tmp + 2 * nb);
        if (rc) return rc;
        hipLaunchKernelGGL((k_scan_apply<P, Op>), dim3((unsigned)(nb - 1)), dim3(256), 0, c->stream, out, n, reverse ? 1 : 0, prefix);
        ZKT_HIP(c, hipGetLastError());
    }
    return ZKT_OK;
}

// ---------------------------------------------------------------------------------------------
// grand products
// ---------------------------------------------------------------------------------------------
template <class P>
__global__ void k_z1_terms(ZTermsArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Fe<P>* num = (Fe<P>*)a.num;
    Fe<P>* den = (Fe<P>*)a.den;
    if (i + 1 == a.n) {  // only n - 1 ratios enter the product (permutation/mod.rs:232)
        fe_store<P>(num + i, fe_one<P>());
        fe_store<P>(den + i, fe_one<P>());
        return;
    }
    const Fe<P> beta = arg_fe<P>(a.beta), gamma = arg_fe<P>(a.gamma);
    const Fe<P> wa = fe_load<P>((const Fe<P>*)a.a + i), wb = fe_load<P>((const Fe<P>*)a.b + i),
                wc = fe_load<P>((const Fe<P>*)a.c + i);
    const Fe<P> br = fe_mul<P>(beta, fe_load<P>((const Fe<P>*)a.roots + i));
    const Fe<P> d2 = fe_dbl<P>(br), d4 = fe_dbl<P>(d2), d8 = fe_dbl<P>(d4);
    const Fe<P> k1br = fe_sub<P>(d8, br);                       // K1 = 7  (permutation/constants.rs:13-15)
    const Fe<P> k2br = fe_add<P>(fe_add<P>(d8, d4), br);        // K2 = 13 (permutation/constants.rs:18-20)
    const Fe<P> ag = fe_add<P>(wa, gamma), bg = fe_add<P>(wb, gamma), cg = fe_add<P>(wc, gamma);
    Fe<P> nu = fe_mul<P>(fe_mul<P>(fe_add<P>(br, ag), fe_add<P>(k1br, bg)), fe_add<P>(k2br, cg));
    Fe<P> de = fe_mul<P>(
        fe_mul<P>(fe_add<P>(fe_mul<P>(beta, fe_load<P>((const Fe<P>*)a.s1 + i)), ag),
                  fe_add<P>(fe_mul<P>(beta, fe_load<P>((const Fe<P>*)a.s2 + i)), bg)),
        fe_add<P>(fe_mul<P>(beta, fe_load<P>((const Fe<P>*)a.s3 + i)), cg));
    fe_store<P>(num + i, nu);
    fe_store<P>(den + i, de);
}

template <class P>
__global__ void k_z2_terms(ZTermsArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Fe<P>* num = (Fe<P>*)a.num;
    Fe<P>* den = (Fe<P>*)a.den;
    if (i + 1 == a.n) {
        fe_store<P>(num + i, fe_one<P>());
        fe_store<P>(den + i, fe_one<P>());
        return;
    }
    const Fe<P> delta = arg_fe<P>(a.delta), eps = arg_fe<P>(a.epsilon);
    const Fe<P> opd = fe_add<P>(fe_one<P>(), delta);
    const Fe<P> eopd = fe_mul<P>(eps, opd);
    const Fe<P>* f = (const Fe<P>*)a.f;
    const Fe<P>* t = (const Fe<P>*)a.t;
    const Fe<P>* h1 = (const Fe<P>*)a.h1;
    const Fe<P>* h2 = (const Fe<P>*)a.h2;
    const Fe<P> ti = fe_load<P>(t + i), tn = fe_load<P>(t + i + 1);
    const Fe<P> h1i = fe_load<P>(h1 + i), h1n = fe_load<P>(h1 + i + 1), h2i = fe_load<P>(h2 + i);
    Fe<P> nu = fe_mul<P>(fe_mul<P>(opd, fe_add<P>(eps, fe_load<P>(f + i))),
                         fe_add<P>(fe_add<P>(fe_mul<P>(delta, tn), eopd), ti));
    Fe<P> de = fe_mul<P>(fe_add<P>(fe_add<P>(fe_mul<P>(delta, h2i), eopd), h1i),
                         fe_add<P>(fe_add<P>(fe_mul<P>(delta, h1n), eopd), h2i));
    fe_store<P>(num + i, nu);
    fe_store<P>(den + i, de);
}

// z[i] = PN_incl[i-1] * SD_incl[i] / prod(den)   (z[0] = 1)
template <class P>
__global__ void k_z_combine(const Fe<P>* pn, const Fe<P>* sd, Fe<P> inv_total, Fe<P>* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> v = fe_mul<P>(fe_load<P>(sd + i), inv_total);
    if (i > 0) v = fe_mul<P>(v, fe_load<P>(pn + i - 1));
    fe_store<P>(out + i, v);
}

// Scalars of a commitment taken in the Lagrange basis (lagrange.hip): the polynomial with evaluations e_i (plus, when
// k > 0, the blinders of prove.rs:472-483 at X^(L+j) and minus themselves at X^j, L = *len its trimmed length) equals
//     sum_{i < n} D[i] S_(i+1) + sum_t D[n + t] V_t,   D[i] = (e_i - e_(i+1)) / n  (e_n := 0),  V_t = [tau^(n+t)] - [tau^t].
// Normally L = n: the blinders are the k extra scalars and the evaluations stay as they are.  If the top coefficients
// happen to vanish (L < n: a constant vector, an empty table) a blinder lands on X^q, q = L + j < n, and changes every
// evaluation by b (w^(iq) - w^(ij)); for q >= n its part beyond X^n goes to the extra scalar q - n and the rest likewise.
template <class P>
__global__ void k_lagrange_scalars(const Fe<P>* ev, size_t n, const uint32_t* len, const Fe<P>* bl, int k, const Fe<P>* roots,
                                   Fe<P> ninv, Fe<P>* D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + (size_t)k) return;
    const size_t L = k ? (size_t)*len : n;
    if (i >= n) {
        const size_t t = i - n;
        Fe<P> acc = fe_zero<P>();
        for (int j = 0; j < k; ++j)
            if (L + j == n + t) acc = fe_add<P>(acc, fe_load<P>(bl + j));
        fe_store<P>(D + i, acc);
        return;
    }
    Fe<P> e0 = fe_load<P>(ev + i), e1 = (i + 1 < n) ? fe_load<P>(ev + i + 1) : fe_zero<P>();
    if (L != n) {
        for (int j = 0; j < k; ++j) {
            const size_t q = (L + j) & (n - 1);
            if (q == (size_t)j) continue;
            const Fe<P> b = fe_load<P>(bl + j);
            e0 = fe_add<P>(e0, fe_mul<P>(b, fe_sub<P>(fe_load<P>(roots + ((i * q) & (n - 1))), fe_load<P>(roots + ((i * j) & (n - 1))))));
            if (i + 1 < n)
                e1 = fe_add<P>(e1, fe_mul<P>(b, fe_sub<P>(fe_load<P>(roots + (((i + 1) * q) & (n - 1))),
                                                           fe_load<P>(roots + (((i + 1) * j) & (n - 1))))));
        }
    }
    fe_store<P>(D + i, fe_mul<P>(fe_sub<P>(e0, e1), ninv));
}

// ---------------------------------------------------------------------------------------------
// quotient: one fused pass over the 4n coset (quotient_poly.rs:98-224)
// ---------------------------------------------------------------------------------------------
// Arithmetic runs on 29-bit limbs (fx.hpp) without leaving them between products.  Two Montgomery scalings
// are in play: values read from memory are arkworks' x R ("A"), host-prepared scalars and the selector
// tables converted by to_hat_form() are x R' = x R 2^5 ("H").  fx_mul(A, H) = A; fx_mul(A, A) = A / 32, and every
// such loss is made good by a power of 32 folded into a scalar (or the q_m table) of the same product.
struct FxArg {
    uint32_t l[9];
};
struct QuotientDev {
    const void *a, *b, *c, *pi, *z1, *z2, *t, *h1, *h2;
    const void *q_m, *q_l, *q_r, *q_o, *q_c, *q_lookup, *q_table, *sigma1, *sigma2, *sigma3, *x, *l1;
    void* out;
    FxArg beta, delta, alpha_k3, alpha2, alpha3_opd_k2, alpha3_k2, alpha4, alpha5;  // H (times 32^k where named _kN)
    FxArg gamma, epsilon, eopd;                                                    // A
    FxArg zh_inv[4];                                                               // H
    uint64_t n4;
    uint64_t first, count;   // the points this launch covers
    const uint32_t* pi_tab;
    uint32_t n_pi_direct;
    uint32_t G, cls, next_off;
    const void *z1_next, *z2_next, *t_next, *h1_next;
};
template <class P>
ZKT_D Fx<P> arg_fx(const FxArg& w) {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = w.l[i];
    return r;
}

template <class P>
__global__ __launch_bounds__(256) void k_quotient(QuotientDev q) {
    static_assert(FxP<P>::L == 9, "scalar field limbs");
    typedef Fx<P> X;
    const uint64_t i = q.first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.first + q.count) return;
    // "omega-next": global index + 4, i.e. entry i + next_off of the (possibly neighbouring) class, wrapping
    const uint64_t j = (i + q.next_off < q.n4) ? i + q.next_off : i + q.next_off - q.n4;
#define LD(ptr, idx) fx_unpack<P>(fe_load<P>((const Fe<P>*)(ptr) + (idx)))   // canonical, < p
    const X beta = arg_fx<P>(q.beta), delta = arg_fx<P>(q.delta), gamma = arg_fx<P>(q.gamma);
    const X a = LD(q.a, i), b = LD(q.b, i), c = LD(q.c, i);
    // keys/arithmetic.rs:67-81 ; q_m is stored as H * 32
    // sums of two products share one reduction (fx_mul2_inl) wherever the two terms have the same scaling
    X acc = fx_mul2_inl<P>(fx_mul<P>(a, b), LD(q.q_m, i), c, LD(q.q_o, i));
    acc = fx_add<P>(acc, fx_mul2_inl<P>(a, LD(q.q_l, i), b, LD(q.q_r, i)));               // < 4p
    const X l1 = LD(q.l1, i);
    if (q.pi_tab) {   // PI on the coset from rotations of l1 (poly.hpp); two terms per reduction, < 16p in all
        X pi = fx_zero<P>();
#pragma unroll 1
        for (uint32_t e = 0; e < q.n_pi_direct; e += 2) {
            X v[2], l[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t* ent = q.pi_tab + (size_t)(e + h) * 10;
                const bool live = e + h < q.n_pi_direct;
                const uint64_t rot = live ? ent[0] : 0;
                const uint64_t at = (i >= rot) ? i - rot : i + q.n4 - rot;
                l[h] = LD(q.l1, at);
#pragma unroll
                for (int w = 0; w < 9; ++w) v[h].l[w] = live ? ent[1 + w] : 0u;
            }
            pi = fx_add<P>(pi, fx_mul2_inl<P>(v[0], l[0], v[1], l[1]));
        }
        acc = fx_add<P>(acc, fx_add<P>(LD(q.q_c, i), pi));                      // < 21p
    } else {
        acc = fx_add<P>(acc, fx_add<P>(LD(q.q_c, i), LD(q.pi, i)));             // < 6p
    }
    // keys/permutation.rs:97-137
    const X z1 = LD(q.z1, i), z1n = LD(q.z1_next, j);
    const X ag = fx_add<P>(a, gamma), bg = fx_add<P>(b, gamma), cg = fx_add<P>(c, gamma);   // < 2p
    const X one = fx_const_to_ark<P>();                                          // 1 in A form
    {
        const X bx = fx_mul<P>(LD(q.x, i), beta);                                // < 2p
        const X d2 = fx_dbl<P>(bx), d4 = fx_dbl<P>(d2), d8 = fx_dbl<P>(d4);      // 4p, 8p, 16p
        const X ak3 = arg_fx<P>(q.alpha_k3);
        X p1 = fx_mul<P>(z1, ak3);
        p1 = fx_mul<P>(p1, fx_add<P>(bx, ag));
        p1 = fx_mul<P>(p1, fx_add<P>(fx_sub<P, 2>(d8, bx), bg));                 // 2p * 20p
        X p2 = fx_mul<P>(z1n, ak3);
        p2 = fx_mul<P>(p2, fx_add<P>(fx_mul<P>(LD(q.sigma1, i), beta), ag));
        p2 = fx_mul<P>(p2, fx_add<P>(fx_mul<P>(LD(q.sigma2, i), beta), bg));
        // p1 * s3 - p2 * t3 with the negated p2 left unnormalised (limbs <= 3 * 2^29): 2p * 28p + 3p * 4p < R' p
        const X p12 = fx_mul2_inl<P>(p1, fx_add<P>(fx_add<P>(fx_add<P>(d8, d4), bx), cg), fx_sub_lazy<P, 3>(fx_zero<P>(), p2),
                                     fx_add<P>(fx_mul<P>(LD(q.sigma3, i), beta), cg));
        acc = fx_add<P>(acc, p12);                                               // < 23p
    }
    {   // keys/lookup.rs:81-122
        const X eps = arg_fx<P>(q.epsilon), eopd = arg_fx<P>(q.eopd);
        const X z2 = LD(q.z2, i), z2n = LD(q.z2_next, j);
        const X t = LD(q.t, i), tn = LD(q.t_next, j), h1 = LD(q.h1, i), h1n = LD(q.h1_next, j), h2 = LD(q.h2, i);
        X k1 = fx_mul<P>(z2, arg_fx<P>(q.alpha3_opd_k2));
        k1 = fx_mul<P>(k1, fx_add<P>(eps, fx_mul<P>(c, LD(q.q_lookup, i))));
        X k2 = fx_mul<P>(z2n, arg_fx<P>(q.alpha3_k2));
        k2 = fx_mul<P>(k2, fx_add<P>(fx_add<P>(eopd, h1), fx_mul<P>(h2, delta)));
        // k1 * U - k2 * V2, same construction as above: 2p * 4p + 3p * 4p
        const X k12 = fx_mul2_inl<P>(k1, fx_add<P>(fx_add<P>(eopd, t), fx_mul<P>(tn, delta)), fx_sub_lazy<P, 3>(fx_zero<P>(), k2),
                                     fx_add<P>(fx_add<P>(eopd, h2), fx_mul<P>(h1n, delta)));
        // alpha^2 (z1 - 1) l1 + alpha^4 (z2 - 1) l1 = l1 * (alpha^2 (z1 - 1) + alpha^4 (z2 - 1))
        const X l13 = fx_mul<P>(l1, fx_mul2_inl<P>(fx_sub<P, 1>(z1, one), arg_fx<P>(q.alpha2), fx_sub<P, 1>(z2, one),
                                                   arg_fx<P>(q.alpha4)));
        const X k4 = fx_mul<P>(fx_mul<P>(t, LD(q.q_table, i)), arg_fx<P>(q.alpha5));
        acc = fx_add<P>(acc, fx_add<P>(fx_add<P>(k12, l13), k4));   // < 29p in all, times zh_inv < p: fits R' p (70 p on BLS12-381)
    }
#undef LD
    // quotient_poly.rs:220-224: times zh_coset[i]^-1; x^n - 1 takes 4 values on the 4n coset (global index mod 4)
    const X r = fx_mul<P>(acc, arg_fx<P>(q.zh_inv[(q.cls + q.G * (uint32_t)i) & 3]));
    fe_store<P>((Fe<P>*)q.out + i, fx_pack<P>(fx_cond_sub_p<P>(r)));
}

// v[i] (A form) -> v[i] * 2^(5 + 5 k32) (H form times 32^k32), canonical packed
template <class P, int K32>
__global__ void k_to_hat(Fe<P>* v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int L = FxP<P>::L;
    constexpr Words<P::N> w = pow2_mod<P>(2 * 29 * L - 32 * P::N + 5 * K32);
    Fx<P> k;
#pragma unroll
    for (int e = 0; e < L; ++e) k.l[e] = FxP<P>::limb_of(w, e);
    fe_store<P>(v + i, fx_pack<P>(fx_cond_sub_p<P>(fx_mul<P>(fx_unpack<P>(fe_load<P>(v + i)), k))));
}

// prove.rs:287-300 after the three chunks were copied out and trimmed (lens[0..2]); lens[3] = len(q)
template <class P>
__global__ void k_quot_blind(Fe<P>* lo, Fe<P>* mid, Fe<P>* hi, const uint32_t* lens, const Fe<P>* b0b1, uint32_t n,
                             uint32_t* status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (lens[3] > 3 * (n + 2)) { atomicOr(status, 2u); return; }       // circuit unsatisfied: degree too high
    if (lens[1] == 0 || lens[2] == 0) { atomicOr(status, 1u); return; }  // reference would panic here
    Fe<P> b0 = fe_load<P>(b0b1), b1 = fe_load<P>(b0b1 + 1);
    fe_store<P>(lo + lens[0], b0);
    fe_store<P>(mid, fe_sub<P>(fe_load<P>(mid), b0));
    fe_store<P>(mid + lens[1], b1);
    fe_store<P>(hi, fe_sub<P>(fe_load<P>(hi), b1));
}

// ---------------------------------------------------------------------------------------------
// opening witness
// ---------------------------------------------------------------------------------------------
constexpr int PW_E = 8;
// The division by (X - z) is a chain of latencies on a stream that waits for it (two scalings around a scan), not a load: the
// workgroups are short (512 elements at the sizes the prover has) so that eight waves per SIMD hide a product's latency, and
// no thread raises z to a power on its own -- the tables come from doubling rounds in one workgroup (k_pow_tables), about 28
// dependent products in all where a per-workgroup z^base cost 30 in every launch.
constexpr int OW_THREADS = 1024;   // k_pow_tables: two halves of 512 (z, 1/z)
// elements per thread of k_mul_pow: 2 unless that needs more than 256 * 256 segment powers (lengths beyond 2^25)
static inline int ow_elems(size_t len) {
    int e = 2;
    while ((len + (size_t)256 * e - 1) / ((size_t)256 * e) > 65536) e <<= 1;
    return e;
}
static inline size_t ow_blocks(size_t len) {
    const size_t seg = (size_t)256 * ow_elems(len);
    return (len + seg - 1) / seg;
}
size_t open_witness_powers(size_t maxlen) { return 2 * (size_t)EV_PW + 2 * ow_blocks(maxlen) + 8; }

// tab[0 .. cnt] = x^0 .. x^cnt: after the round with stride s the entries up to 2 s exist (tab[s + j] = tab[s] tab[j], j = 1..s).
// Called by the whole workgroup with a uniform cnt; lt = index within the half that owns `tab`.
template <class P>
ZKT_D void pow_rounds(Fe<P>* tab, const Fe<P>& x, int cnt, int lt) {
    if (lt == 0) {
        tab[0] = fe_one<P>();
        tab[1] = x;
    }
    __syncthreads();
    for (int s = 1; s < cnt; s <<= 1) {
        for (int j = lt + 1; j <= s && s + j <= cnt; j += OW_THREADS / 2) tab[s + j] = fe_mul<P>(tab[s], tab[j]);
        __syncthreads();
    }
}
// pw[0][t] = z^t, pw[1][t] = zinv^t, t = 0..256 (two rows of EV_PW), then blk[0][b] = z^(256 E b), blk[1][b] = zinv^(256 E b), b < nblk
template <class P>
__global__ __launch_bounds__(OW_THREADS) void k_pow_tables(Fe<P> z, Fe<P> zinv, Fe<P>* pw, int nblk, int E) {
    __shared__ Fe<P> row[2][EV_PW], seg[2][EV_PW], top[2][EV_PW];
    const int half = threadIdx.x / (OW_THREADS / 2), lt = threadIdx.x % (OW_THREADS / 2);
    pow_rounds<P>(row[half], half ? zinv : z, 256, lt);
    Fe<P> y = row[half][256];                       // x^(256 E)
    for (int e = 1; e < E; e <<= 1) y = fe_sqr<P>(y);
    pow_rounds<P>(seg[half], y, 256, lt);
    const int ntop = (nblk + 255) / 256;            // <= 256 (ow_elems)
    pow_rounds<P>(top[half], seg[half][256], ntop, lt);
    for (int t = lt; t < EV_PW; t += OW_THREADS / 2) fe_store<P>(pw + half * EV_PW + t, row[half][t]);
    Fe<P>* blk = pw + 2 * EV_PW + (size_t)half * nblk;
    for (int b = lt; b < nblk; b += OW_THREADS / 2) fe_store<P>(blk + b, fe_mul<P>(seg[half][b & 255], top[half][b >> 8]));
}
// out[i] = in[i] * x^(i + shift) for i < n, zero for n <= i < cap (shift 0 or 1).  One workgroup per 256 E consecutive elements,
// thread t takes i = base + 256 j + t (coalesced): x^(i + shift) = blk[workgroup] * pw[t + shift] * (x^256)^j.
template <class P>
__global__ __launch_bounds__(256) void k_mul_pow(const Fe<P>* in, Fe<P>* out, size_t n, size_t cap, const Fe<P>* pw, const Fe<P>* blk,
                                                 int E, int shift) {
    const size_t base = (size_t)blockIdx.x * 256 * E;
    Fe<P> r = fe_mul<P>(fe_load<P>(blk + blockIdx.x), fe_load<P>(pw + threadIdx.x + shift));
#pragma unroll 1
    for (int j = 0; j < E; ++j) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        if (i >= cap) break;
        fe_store<P>(out + i, i < n ? fe_mul<P>(fe_load<P>(in + i), r) : fe_zero<P>());
        if (j + 1 < E) r = fe_mul<P>(r, fe_load<P>(pw + 256));
    }
}

template <class P>
__global__ void k_gen_powers(Fe<P>* out, size_t n, Fe<P> base, Fe<P> scale) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * PW_E;
    if (i0 >= n) return;
    Fe<P> pw = fe_mul<P>(scale, fe_pow_u64<P>(base, i0));
#pragma unroll 1
    for (int e = 0; e < PW_E && i0 + e < n; ++e) {
        fe_store<P>(out + i0 + e, pw);
        pw = fe_mul<P>(pw, base);
    }
}

// ---------------------------------------------------------------------------------------------
// Plookup h1 / h2 (lookup/multiset.rs:103-146): counts per table key, then run-length expansion
// ---------------------------------------------------------------------------------------------
template <class P>
ZKT_D int key_cmp(const Fe<P>& a, const Fe<P>& b) {  // order on the Montgomery words (any total order works)
#pragma unroll
    for (int i = P::N - 1; i >= 0; --i) {
        if (a.v[i] < b.v[i]) return -1;
        if (a.v[i] > b.v[i]) return 1;
    }
    return 0;
}

constexpr int LK_LDS_KEYS = 8192;
template <class P>
__global__ __launch_bounds__(256) void k_lookup_count(const Fe<P>* f, size_t n, const Fe<P>* sorted_keys,
                                                      const uint32_t* perm, uint32_t nkeys, uint32_t* counts,
                                                      uint32_t* status) {
    __shared__ uint32_t hist[LK_LDS_KEYS];
    const bool use_lds = nkeys <= LK_LDS_KEYS;
    if (use_lds) {
        for (uint32_t k = threadIdx.x; k < nkeys; k += blockDim.x) hist[k] = 0;
        __syncthreads();
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Fe<P> x = fe_load<P>(f + i);
        uint32_t lo = 0, hi = nkeys;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (key_cmp<P>(fe_load<P>(sorted_keys + mid), x) < 0) lo = mid + 1; else hi = mid;
        }
        if (lo >= nkeys || key_cmp<P>(fe_load<P>(sorted_keys + lo), x) != 0) {
            atomicOr(status, 4u);  // Error::ElementNotIndexedInTable
            continue;
        }
        const uint32_t k = perm[lo];
        if (use_lds) atomicAdd(&hist[k], 1u); else atomicAdd(counts + k, 1u);
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nkeys; k += blockDim.x)
            if (hist[k]) atomicAdd(counts + k, hist[k]);
    }
}

// multiset.rs:126-143 on the device: every key contributes count/2 copies to both halves and, when its count is
// odd, one more copy alternately to the even half and the odd half.  The alternation is a parity prefix, so the
// start offsets of both halves are three exclusive scans.  One workgroup of 1024; status |= 8 when a half does not
// end at n.
// counts[k] = base[k] (multiplicity in the padded table) + hits[k] (occurrences in f); hits is cleared on the way so
// that the next proof finds it zero without a memset.
__global__ __launch_bounds__(1024) void k_lookup_starts(const uint32_t* base_counts, uint32_t* hits, uint32_t nk, uint32_t n,
                                                        uint32_t* s_even, uint32_t* s_odd, uint32_t* status) {
    __shared__ uint32_t wsum[3][16];
    uint32_t carry[3] = {0, 0, 0};   // odd keys so far, even-half length, odd-half length
    for (uint32_t base = 0; base < nk; base += 1024) {
        const uint32_t k = base + threadIdx.x;
        uint32_t cnt = 0;
        if (k < nk) {
            cnt = base_counts[k] + hits[k];
            hits[k] = 0;
        }
        uint32_t v[3] = {cnt & 1u, 0u, 0u};
        uint32_t ex[3], tot[3];
#pragma unroll 1
        for (int q = 0; q < 3; ++q) {
            if (q == 1) {   // the parity prefix is known now
                const uint32_t odd_before = carry[0] + ex[0];
                const uint32_t to_even = (cnt & 1u) && ((odd_before & 1u) == 0);
                v[1] = (cnt >> 1) + (to_even ? 1u : 0u);
                v[2] = (cnt >> 1) + (((cnt & 1u) && !to_even) ? 1u : 0u);
            }
            uint32_t incl = v[q];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)(threadIdx.x & 63) >= d) incl += o;
            }
            if ((threadIdx.x & 63) == 63) wsum[q][threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
            for (uint32_t w = 0; w < 16; ++w) {
                const uint32_t x = wsum[q][w];
                if (w < (threadIdx.x >> 6)) before += x;
                all += x;
            }
            ex[q] = before + incl - v[q];
            tot[q] = all;
        }
        if (k < nk) {
            s_even[k] = carry[1] + ex[1];
            s_odd[k] = carry[2] + ex[2];
        }
        __syncthreads();   // wsum is rewritten by the next chunk
        for (int q = 0; q < 3; ++q) carry[q] += tot[q];
    }
    if (threadIdx.x == 0) {
        s_even[nk] = carry[1];
        s_odd[nk] = carry[2];
        if (carry[1] != n || carry[2] != n) atomicOr(status, 8u);
    }
}

// out[p] = key k with starts[k] <= p < starts[k+1]
template <class P>
__global__ void k_lookup_expand(const Fe<P>* keys, const uint32_t* starts, uint32_t nkeys, Fe<P>* out, size_t n) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t lo = 0, hi = nkeys;  // largest k with starts[k] <= p
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (starts[mid] <= p) lo = mid; else hi = mid;
    }
    fe_store<P>(out + p, fe_load<P>(keys + lo));
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (dispatch on the context's scalar field)
// ---------------------------------------------------------------------------------------------
#define ZKT_DISPATCH(c, FN, ...)                                                  \
    do {                                                                          \
        if ((c)->curve == ZKT_CURVE_BN254) return FN<Bn254Fr>(c, __VA_ARGS__);    \
        return FN<Bls381Fr>(c, __VA_ARGS__);                                      \
    } while (0)

static inline unsigned nblocks(size_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

template <class P>
static Fe<P> host_fe(const uint32_t* w) {
    Fe<P> r;
    for (int i = 0; i < 8; ++i) r.v[i] = w[i];
    return r;
}

// out[0..n) = in[0..len) followed by zeros (prove.rs:39-55 pad_to)
template <class P>
__global__ void k_copy_pad(const Fe<P>* in, size_t len, Fe<P>* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store<P>(out + i, i < len ? fe_load<P>(in + i) : fe_zero<P>());
}
// prove.rs:49-55 wire_evals + pad_to: out[i] = values[idx[i]] for i < rows (0 for Variable::Zero), zero above
template <class P>
__global__ void k_gather_pad(const Fe<P>* values, uint32_t n_vars, const uint32_t* idx, size_t rows, Fe<P>* out, size_t n,
                             uint32_t* status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> v = fe_zero<P>();
    if (i < rows) {
        const uint32_t k = idx[i];
        if (k != 0xFFFFFFFFu) {
            if (k < n_vars) v = fe_load<P>(values + k);
            else atomicOr(status, 16u);
        }
    }
    fe_store<P>(out + i, v);
}
template <class P> static int gather_pad_t(zkt_ctx* c, const void* values, size_t n_vars, const uint32_t* idx, size_t rows,
                                           void* out, size_t n, uint32_t* status) {
    if (!n) return ZKT_OK;
    hipLaunchKernelGGL(k_gather_pad<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Fe<P>*)values, (uint32_t)n_vars, idx,
                       rows, (Fe<P>*)out, n, status);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_gather_pad(zkt_ctx* c, const void* d_values, size_t n_vars, const uint32_t* d_idx, size_t rows, void* out, size_t n,
                    uint32_t* d_status) {
    ZKT_DISPATCH(c, gather_pad_t, d_values, n_vars, d_idx, rows, out, n, d_status);
}

template <class P> static int copy_pad_t(zkt_ctx* c, const void* in, size_t len, void* out, size_t n) {
    if (!n) return ZKT_OK;
    hipLaunchKernelGGL(k_copy_pad<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Fe<P>*)in, len, (Fe<P>*)out, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_copy_pad(zkt_ctx* c, const void* d_in, size_t len, void* out, size_t n) { ZKT_DISPATCH(c, copy_pad_t, d_in, len, out, n); }

template <class P> static int mul_vec_t(zkt_ctx* c, const void* a, const void* b, void* o, size_t n) {
    if (!n) return ZKT_OK;
    hipLaunchKernelGGL(k_mul_vec<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Fe<P>*)a, (const Fe<P>*)b, (Fe<P>*)o, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_mul_vec(zkt_ctx* c, const void* a, const void* b, void* out, size_t n) { ZKT_DISPATCH(c, mul_vec_t, a, b, out, n); }

int poly_set_zero(zkt_ctx* c, void* p, size_t n_elems) {
    if (n_elems) ZKT_HIP(c, hipMemsetAsync(p, 0, n_elems * 32, c->stream));
    return ZKT_OK;
}

// *d_len must be zero on entry (the prover clears all its length slots with the status words at proof start)
template <class P> static int trim_len_t(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len, void* zero_at = nullptr,
                                         int zero_count = 0) {
    if (!n) return ZKT_OK;
    const size_t tail = n > 1024 ? n - 1024 : 0;
    hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(n - tail)), dim3(256), 0, c->stream, (const Fe<P>*)p, tail, n, d_len, 0,
                       (Fe<P>*)zero_at, zero_count);
    ZKT_HIP(c, hipGetLastError());
    if (tail) {  // full scan, a no-op unless the top 1024 coefficients were all zero
        hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(tail)), dim3(256), 0, c->stream, (const Fe<P>*)p, (size_t)0, tail, d_len, 1,
                           (Fe<P>*)nullptr, 0);
        ZKT_HIP(c, hipGetLastError());
    }
    return ZKT_OK;
}
template <class P> static int trim_len_pub_t(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len, void* zero_at, int zero_count) {
    return trim_len_t<P>(c, p, n, d_len, zero_at, zero_count);
}
int poly_trim_len(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len, void* zero_at, int zero_count) {
    ZKT_DISPATCH(c, trim_len_pub_t, p, n, d_len, zero_at, zero_count);
}

template <class P> static int add_blinders_t(zkt_ctx* c, void* p, const uint32_t* d_len, const void* bl, int k) {
    hipLaunchKernelGGL(k_add_blinders<P>, dim3(1), dim3(64), 0, c->stream, (Fe<P>*)p, d_len, (const Fe<P>*)bl, k);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_add_blinders(zkt_ctx* c, void* p, const uint32_t* d_len, const void* d_blinders, int k, size_t) {
    ZKT_DISPATCH(c, add_blinders_t, p, d_len, d_blinders, k);
}

template <class P> static int lincomb_t(zkt_ctx* c, const LinCombArgs& a, void* out, size_t n) {
    static_assert(LC_MAX_TERMS <= 32, "the lazy sum of the terms must stay below 2^6 p");
    LinCombDev d{};
    d.nterms = a.nterms;
    for (int k = 0; k < a.nterms; ++k) {
        d.poly[k] = a.poly[k];
        d.len[k] = a.len[k];
        const Fx<P> h = fx_cond_sub_p<P>(fx_from_ark<P>(host_fe<P>(a.scalar[k])));
        for (int w = 0; w < 9; ++w) d.scalar[k][w] = h.l[w];
    }
    hipLaunchKernelGGL(k_lincomb<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, d, (Fe<P>*)out, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_lincomb(zkt_ctx* c, const LinCombArgs& a, void* out, size_t n) { ZKT_DISPATCH(c, lincomb_t, a, out, n); }

template <class P> static int eval_many_t(zkt_ctx* c, const EvalArgs& a, void* d_partials, void* d_results, void* d_powers) {
    uint64_t maxlen = 1;
    for (int k = 0; k < a.count; ++k) if (a.len[k] > maxlen) maxlen = a.len[k];
    int nblk = (int)((maxlen + EV_SEG - 1) / EV_SEG);
    hipLaunchKernelGGL(k_eval_powers<P>, dim3(a.count), dim3(1024), 0, c->stream, a, (Fe<P>*)d_powers, nblk);
    hipLaunchKernelGGL(k_eval_partial<P>, dim3(nblk, a.count), dim3(256), 0, c->stream, a, (const Fe<P>*)d_powers,
                       (Fe<P>*)d_partials, nblk);
    ZKT_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_eval_final<P>, dim3(a.count), dim3(256), 0, c->stream, (const Fe<P>*)d_partials, nblk, (Fe<P>*)d_results);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_eval_many(zkt_ctx* c, const EvalArgs& a, void* d_partials, void* d_results, void* d_powers) {
    ZKT_DISPATCH(c, eval_many_t, a, d_partials, d_results, d_powers);
}

template <class P> static int z1_terms_t(zkt_ctx* c, const ZTermsArgs& a) {
    hipLaunchKernelGGL(k_z1_terms<P>, dim3(nblocks(a.n)), dim3(256), 0, c->stream, a);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int z1_terms(zkt_ctx* c, const ZTermsArgs& a) { ZKT_DISPATCH(c, z1_terms_t, a); }
template <class P> static int z2_terms_t(zkt_ctx* c, const ZTermsArgs& a) {
    hipLaunchKernelGGL(k_z2_terms<P>, dim3(nblocks(a.n)), dim3(256), 0, c->stream, a);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int z2_terms(zkt_ctx* c, const ZTermsArgs& a) { ZKT_DISPATCH(c, z2_terms_t, a); }

template <class P> static int scan_mul_t(zkt_ctx* c, const void* in, void* out, size_t n, bool rev, void* tmp) {
    return scan_t<P, OpMul>(c, (const Fe<P>*)in, (Fe<P>*)out, n, rev, (Fe<P>*)tmp);
}
int scan_mul(zkt_ctx* c, const void* in, void* out, size_t n, bool reverse, void* d_tmp) { ZKT_DISPATCH(c, scan_mul_t, in, out, n, reverse, d_tmp); }
template <class P> static int scan_add_t(zkt_ctx* c, const void* in, void* out, size_t n, bool rev, void* tmp) {
    return scan_t<P, OpAdd>(c, (const Fe<P>*)in, (Fe<P>*)out, n, rev, (Fe<P>*)tmp);
}
int scan_add(zkt_ctx* c, const void* in, void* out, size_t n, bool reverse, void* d_tmp) { ZKT_DISPATCH(c, scan_add_t, in, out, n, reverse, d_tmp); }

template <class P> static int z_combine_t(zkt_ctx* c, const void* pn, const void* sd, const uint32_t* inv, void* out, size_t n) {
    hipLaunchKernelGGL(k_z_combine<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Fe<P>*)pn, (const Fe<P>*)sd, host_fe<P>(inv), (Fe<P>*)out, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int z_combine(zkt_ctx* c, const void* pn, const void* sd, const uint32_t inv_total[8], void* out, size_t n) { ZKT_DISPATCH(c, z_combine_t, pn, sd, inv_total, out, n); }

template <class P> static int lagrange_scalars_t(zkt_ctx* c, const void* ev, size_t n, const uint32_t* d_len, const void* d_bl, int k,
                                                 const void* roots, void* out) {
    Fe<P> nn = fe_zero<P>();
    nn.v[0] = (uint32_t)(n & 0xffffffffu);
    nn.v[1] = (uint32_t)((uint64_t)n >> 32);
    const Fe<P> ninv = fe_inv_host<P>(fe_to_mont<P>(nn));
    hipLaunchKernelGGL(k_lagrange_scalars<P>, dim3(nblocks(n + (size_t)k)), dim3(256), 0, c->stream, (const Fe<P>*)ev, n, d_len,
                       (const Fe<P>*)d_bl, k, (const Fe<P>*)roots, ninv, (Fe<P>*)out);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int lagrange_scalars(zkt_ctx* c, const void* ev, size_t n, const uint32_t* d_len, const void* d_blinders, int k, const void* d_roots,
                     void* out) {
    ZKT_DISPATCH(c, lagrange_scalars_t, ev, n, d_len, d_blinders, k, d_roots, out);
}

template <class P> static FxArg hat_arg(const Fe<P>& s) {   // A form -> canonical H-form limbs
    const Fx<P> h = fx_cond_sub_p<P>(fx_from_ark<P>(s));
    FxArg r;
    for (int i = 0; i < 9; ++i) r.l[i] = h.l[i];
    return r;
}
template <class P> static FxArg ark_arg(const Fe<P>& s) {
    const Fx<P> h = fx_unpack<P>(s);
    FxArg r;
    for (int i = 0; i < 9; ++i) r.l[i] = h.l[i];
    return r;
}
template <class P> static int quotient_t(zkt_ctx* c, const QuotientArgs& a) {
    ProfScope prof(c, "quotient");
    auto get = [](const uint32_t* w) { Fe<P> r; for (int i = 0; i < 8; ++i) r.v[i] = w[i]; return r; };
    const Fe<P> alpha = get(a.alpha), beta = get(a.beta), gamma = get(a.gamma), delta = get(a.delta), eps = get(a.epsilon);
    const Fe<P> k2 = fe_from_u32<P>(1u << 10), k3 = fe_from_u32<P>(1u << 15);
    const Fe<P> a2 = fe_sqr<P>(alpha), a3 = fe_mul<P>(a2, alpha), a4 = fe_mul<P>(a3, alpha), a5 = fe_mul<P>(a4, alpha);
    const Fe<P> opd = fe_add<P>(delta, fe_one<P>());
    QuotientDev q{};
    q.a = a.a; q.b = a.b; q.c = a.c; q.pi = a.pi; q.z1 = a.z1; q.z2 = a.z2; q.t = a.t; q.h1 = a.h1; q.h2 = a.h2;
    q.q_m = a.q_m; q.q_l = a.q_l; q.q_r = a.q_r; q.q_o = a.q_o; q.q_c = a.q_c; q.q_lookup = a.q_lookup; q.q_table = a.q_table;
    q.sigma1 = a.sigma1; q.sigma2 = a.sigma2; q.sigma3 = a.sigma3; q.x = a.x; q.l1 = a.l1; q.out = a.out;
    q.beta = hat_arg<P>(beta);
    q.delta = hat_arg<P>(delta);
    q.alpha_k3 = hat_arg<P>(fe_mul<P>(alpha, k3));
    q.alpha2 = hat_arg<P>(a2);
    q.alpha3_opd_k2 = hat_arg<P>(fe_mul<P>(fe_mul<P>(a3, opd), k2));
    q.alpha3_k2 = hat_arg<P>(fe_mul<P>(a3, k2));
    q.alpha4 = hat_arg<P>(a4);
    q.alpha5 = hat_arg<P>(a5);
    q.gamma = ark_arg<P>(gamma);
    q.epsilon = ark_arg<P>(eps);
    q.eopd = ark_arg<P>(fe_mul<P>(eps, opd));
    for (int k = 0; k < 4; ++k) q.zh_inv[k] = hat_arg<P>(get(a.zh_inv[k]));
    q.n4 = a.n4;
    q.pi_tab = a.pi_tab;
    q.n_pi_direct = a.n_pi_direct;
    if (a.G <= 1) {
        q.G = 1; q.cls = 0; q.next_off = 4;
        q.z1_next = a.z1; q.z2_next = a.z2; q.t_next = a.t; q.h1_next = a.h1;
    } else {
        q.G = a.G; q.cls = a.cls; q.next_off = a.next_off;
        q.z1_next = a.z1_next; q.z2_next = a.z2_next; q.t_next = a.t_next; q.h1_next = a.h1_next;
    }
    if (a.pi_tab && a.n_pi_direct > (uint32_t)QUOTIENT_PI_DIRECT_MAX)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "too many direct public inputs");
    q.first = a.first;
    q.count = a.count ? a.count : a.n4 - a.first;
    if (q.first + q.count > a.n4) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "quotient range outside the vectors");
    hipLaunchKernelGGL(k_quotient<P>, dim3(nblocks(q.count)), dim3(256), 0, c->stream, q);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
template <class P> static int to_hat_t(zkt_ctx* c, void* v, size_t n, int k32) {
    if (k32 == 0) hipLaunchKernelGGL((k_to_hat<P, 0>), dim3(nblocks(n)), dim3(256), 0, c->stream, (Fe<P>*)v, n);
    else hipLaunchKernelGGL((k_to_hat<P, 1>), dim3(nblocks(n)), dim3(256), 0, c->stream, (Fe<P>*)v, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int to_hat_form(zkt_ctx* c, void* v, size_t n, int k32) { ZKT_DISPATCH(c, to_hat_t, v, n, k32); }
int quotient_pointwise(zkt_ctx* c, const QuotientArgs& a) { ZKT_DISPATCH(c, quotient_t, a); }

// after the all-gather of a sharded proof: class-major -> natural order of the 4n coset
template <class P>
__global__ void k_interleave(const Fe<P>* in, Fe<P>* out, uint64_t n4, uint32_t log_g, uint64_t piece) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // natural index: coalesced writes
    if (t >= n4) return;
    const uint64_t cls = t & ((1u << log_g) - 1), i = t >> log_g;
    // `in` is [chunk][class][piece]: entry i of class cls sits in chunk i / piece
    const uint64_t j = i / piece, o = i - j * piece;
    fe_store<P>(out + t, fe_load<P>(in + ((j << log_g) + cls) * piece + o));
}
template <class P> static int interleave_t(zkt_ctx* c, const void* in, void* out, size_t n4, uint32_t G, uint32_t chunks) {
    uint32_t lg = 0;
    while ((1u << lg) < G) ++lg;
    const uint64_t m = n4 >> lg;
    if (chunks == 0 || m % chunks) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "interleave: pieces must divide the class");
    hipLaunchKernelGGL(k_interleave<P>, dim3(nblocks(n4)), dim3(256), 0, c->stream, (const Fe<P>*)in, (Fe<P>*)out, (uint64_t)n4, lg,
                       m / chunks);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int quotient_interleave(zkt_ctx* c, const void* in, void* out, size_t n4, uint32_t G, uint32_t chunks) {
    ZKT_DISPATCH(c, interleave_t, in, out, n4, G, chunks);
}

// prove.rs:287-289: the three (n+2)-coefficient chunks of the quotient, each zero-padded to its buffer
template <class P>
__global__ void k_quot_split(const Fe<P>* q, size_t chunk, size_t cap, Fe<P>* lo, Fe<P>* mid, Fe<P>* hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const bool in = i < chunk;
    const Fe<P> z = fe_zero<P>();
    fe_store<P>(lo + i, in ? fe_load<P>(q + i) : z);
    fe_store<P>(mid + i, in ? fe_load<P>(q + chunk + i) : z);
    fe_store<P>(hi + i, in ? fe_load<P>(q + 2 * chunk + i) : z);
}

template <class P> static int quot_split_t(zkt_ctx* c, const void* q, size_t n, const void* b0b1, void* lo, void* mid, void* hi, uint32_t* d_status) {
    // d_status[0] = error bits ; d_status[4..7] = lens (lo, mid, hi, q), zero on entry
    const Fe<P>* Q = (const Fe<P>*)q;
    const size_t cap = n + 8, chunk = n + 2;
    hipLaunchKernelGGL(k_quot_split<P>, dim3(nblocks(cap)), dim3(256), 0, c->stream, Q, chunk, cap, (Fe<P>*)lo, (Fe<P>*)mid,
                       (Fe<P>*)hi);
    ZKT_HIP(c, hipGetLastError());
    uint32_t* lens = d_status + 4;
    int rc;
    if ((rc = trim_len_t<P>(c, lo, chunk, lens + 0))) return rc;
    if ((rc = trim_len_t<P>(c, mid, chunk, lens + 1))) return rc;
    if ((rc = trim_len_t<P>(c, hi, chunk, lens + 2))) return rc;
    {   // only "is anything non-zero at or above 3(n+2)?" matters for the quotient itself
        const size_t lo = 3 * chunk, hi = 4 * n;
        hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(hi - lo)), dim3(256), 0, c->stream, (const Fe<P>*)q, lo, hi, lens + 3, 0,
                           (Fe<P>*)nullptr, 0);
        ZKT_HIP(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_quot_blind<P>, dim3(1), dim3(64), 0, c->stream, (Fe<P>*)lo, (Fe<P>*)mid, (Fe<P>*)hi, lens, (const Fe<P>*)b0b1, (uint32_t)n, d_status);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int quotient_split_blind(zkt_ctx* c, const void* q, size_t n, const void* d_b0b1, void* q_lo, void* q_mid, void* q_hi, uint32_t* d_status) {
    ZKT_DISPATCH(c, quot_split_t, q, n, d_b0b1, q_lo, q_mid, q_hi, d_status);
}

template <class P> static int open_witness_t(zkt_ctx* c, const void* p, size_t len, const uint32_t* z, const uint32_t* zinv, void* ta, void* tb, void* scan_tmp, void* out, void* d_powers) {
    if (len == 0) return ZKT_OK;
    // w_j = z^-(j+1) * sum_{i > j} p_i z^i: scale by z^i, suffix sums, scale by z^-(j+1)
    const Fe<P> zz = host_fe<P>(z), zi = host_fe<P>(zinv);
    Fe<P>* pw = (Fe<P>*)d_powers;
    const int E = ow_elems(len);
    const unsigned blocks = (unsigned)ow_blocks(len);
    const Fe<P>* blk = pw + 2 * EV_PW;
    hipLaunchKernelGGL(k_pow_tables<P>, dim3(1), dim3(OW_THREADS), 0, c->stream, zz, zi, pw, (int)blocks, E);
    hipLaunchKernelGGL(k_mul_pow<P>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<P>*)p, (Fe<P>*)ta, len, len,
                       (const Fe<P>*)pw, blk, E, 0);
    ZKT_HIP(c, hipGetLastError());
    int rc = scan_t<P, OpAdd>(c, (const Fe<P>*)ta, (Fe<P>*)tb, len, true, (Fe<P>*)scan_tmp);
    if (rc) return rc;
    // out[j] = S[j + 1] * zinv^(j + 1) for j + 1 < len, zero at j = len - 1
    hipLaunchKernelGGL(k_mul_pow<P>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<P>*)tb + 1, (Fe<P>*)out, len - 1, len,
                       (const Fe<P>*)pw + EV_PW, blk + blocks, E, 1);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int open_witness(zkt_ctx* c, const void* p, size_t len, const uint32_t z[8], const uint32_t z_inv[8], void* d_tmp_a, void* d_tmp_b, void* d_scan_tmp, void* out, void* d_powers) {
    ZKT_DISPATCH(c, open_witness_t, p, len, z, z_inv, d_tmp_a, d_tmp_b, d_scan_tmp, out, d_powers);
}

template <class P> static int gen_powers_t(zkt_ctx* c, void* out, size_t n, const uint32_t* base, const uint32_t* scale) {
    if (!n) return ZKT_OK;
    hipLaunchKernelGGL(k_gen_powers<P>, dim3(nblocks((n + PW_E - 1) / PW_E)), dim3(256), 0, c->stream, (Fe<P>*)out, n, host_fe<P>(base), host_fe<P>(scale));
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int gen_powers(zkt_ctx* c, void* out, size_t n, const uint32_t base[8], const uint32_t scale[8]) { ZKT_DISPATCH(c, gen_powers_t, out, n, base, scale); }

template <class P> static int lookup_count_t(zkt_ctx* c, const void* f, size_t n, const void* keys, const uint32_t* perm, uint32_t nkeys, uint32_t* counts, uint32_t* status) {
    unsigned blocks = nblocks(n);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_lookup_count<P>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<P>*)f, n, (const Fe<P>*)keys, perm, nkeys, counts, status);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int lookup_count(zkt_ctx* c, const void* f, size_t n, const void* d_sorted_keys, const uint32_t* d_perm, uint32_t nkeys, uint32_t* d_counts, uint32_t* d_status) {
    ZKT_DISPATCH(c, lookup_count_t, f, n, d_sorted_keys, d_perm, nkeys, d_counts, d_status);
}
template <class P> static int lookup_expand_t(zkt_ctx* c, const void* keys, const uint32_t* starts, uint32_t nkeys, void* out, size_t n) {
    hipLaunchKernelGGL(k_lookup_expand<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, (const Fe<P>*)keys, starts, nkeys, (Fe<P>*)out, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int lookup_starts(zkt_ctx* c, const uint32_t* d_base_counts, uint32_t* d_hits, uint32_t nkeys, size_t n, uint32_t* d_even,
                  uint32_t* d_odd, uint32_t* d_status) {
    hipLaunchKernelGGL(k_lookup_starts, dim3(1), dim3(1024), 0, c->stream, d_base_counts, d_hits, nkeys, (uint32_t)n, d_even,
                       d_odd, d_status);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int lookup_expand(zkt_ctx* c, const void* d_keys_insertion, const uint32_t* d_starts, uint32_t nkeys, void* out, size_t n) {
    ZKT_DISPATCH(c, lookup_expand_t, d_keys_insertion, d_starts, nkeys, out, n);
}

}  // namespace zkt
