// Polynomial / evaluation-vector kernels of the prover rounds (see poly.hpp for the reference
// lines each one replaces).  All values are Fr in Montgomery form, arkworks layout.
#include "poly.hpp"

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
__global__ void k_trim_len(const Fe<P>* p, size_t lo, size_t n, uint32_t* len, int skip_if_set) {
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

template <class P>
__global__ void k_lincomb(LinCombArgs a, Fe<P>* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> acc = fe_zero<P>();
    for (int k = 0; k < a.nterms; ++k) {
        if (i < a.len[k]) {
            Fe<P> s = arg_fe<P>(a.scalar[k]);
            acc = fe_add<P>(acc, fe_mul<P>(s, fe_load<P>((const Fe<P>*)a.poly[k] + i)));
        }
    }
    fe_store<P>(out + i, acc);
}

// ---------------------------------------------------------------------------------------------
// batched point evaluation: result[k] = poly[k](point[k])
// ---------------------------------------------------------------------------------------------
constexpr int EV_E = 8;
constexpr int EV_SEG = 256 * EV_E;

template <class P>
__global__ __launch_bounds__(256) void k_eval_partial(EvalArgs a, Fe<P>* partials, int nblk) {
    __shared__ Fe<P> red[256];
    const int k = blockIdx.y;
    const Fe<P>* poly = (const Fe<P>*)a.poly[k];
    const uint64_t len = a.len[k];
    const Fe<P> x = arg_fe<P>(a.point[k]);
    const uint64_t base = (uint64_t)blockIdx.x * EV_SEG;
    const int t = threadIdx.x;
    Fe<P> acc = fe_zero<P>();
    const uint64_t i0 = base + (uint64_t)t * EV_E;
    if (i0 < len) {
#pragma unroll 1
        for (int j = EV_E - 1; j >= 0; --j) {
            acc = fe_mul<P>(acc, x);
            if (i0 + j < len) acc = fe_add<P>(acc, fe_load<P>(poly + i0 + j));
        }
        acc = fe_mul<P>(acc, fe_pow_u64<P>(x, (uint64_t)t * EV_E));
    }
    red[t] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) red[t] = fe_add<P>(red[t], red[t + d]);
        __syncthreads();
    }
    if (t == 0) fe_store<P>(partials + (size_t)k * nblk + blockIdx.x, fe_mul<P>(red[0], fe_pow_u64<P>(x, base)));
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

// ---------------------------------------------------------------------------------------------
// quotient: one fused pass over the 4n coset (quotient_poly.rs:98-224)
// ---------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void k_quotient(QuotientArgs q) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.n4) return;
    const uint64_t j = (i + 4 < q.n4) ? i + 4 : i + 4 - q.n4;  // "omega-next" on the 4n coset
#define LD(ptr, idx) fe_load<P>((const Fe<P>*)(ptr) + (idx))
    const Fe<P> alpha = arg_fe<P>(q.alpha), beta = arg_fe<P>(q.beta), gamma = arg_fe<P>(q.gamma),
                delta = arg_fe<P>(q.delta), eps = arg_fe<P>(q.epsilon);
    const Fe<P> a = LD(q.a, i), b = LD(q.b, i), c = LD(q.c, i);
    // keys/arithmetic.rs:67-81
    Fe<P> acc = fe_mul<P>(fe_mul<P>(a, b), LD(q.q_m, i));
    acc = fe_add<P>(acc, fe_mul<P>(a, LD(q.q_l, i)));
    acc = fe_add<P>(acc, fe_mul<P>(b, LD(q.q_r, i)));
    acc = fe_add<P>(acc, fe_mul<P>(c, LD(q.q_o, i)));
    acc = fe_add<P>(acc, LD(q.q_c, i));
    acc = fe_add<P>(acc, LD(q.pi, i));
    // keys/permutation.rs:97-137
    const Fe<P> z1 = LD(q.z1, i), z1n = LD(q.z1, j), l1 = LD(q.l1, i);
    const Fe<P> ag = fe_add<P>(a, gamma), bg = fe_add<P>(b, gamma), cg = fe_add<P>(c, gamma);
    {
        const Fe<P> bx = fe_mul<P>(beta, LD(q.x, i));
        const Fe<P> d2 = fe_dbl<P>(bx), d4 = fe_dbl<P>(d2), d8 = fe_dbl<P>(d4);
        Fe<P> p1 = fe_mul<P>(alpha, z1);
        p1 = fe_mul<P>(p1, fe_add<P>(bx, ag));
        p1 = fe_mul<P>(p1, fe_add<P>(fe_sub<P>(d8, bx), bg));
        p1 = fe_mul<P>(p1, fe_add<P>(fe_add<P>(fe_add<P>(d8, d4), bx), cg));
        Fe<P> p2 = fe_mul<P>(alpha, z1n);
        p2 = fe_mul<P>(p2, fe_add<P>(fe_mul<P>(beta, LD(q.sigma1, i)), ag));
        p2 = fe_mul<P>(p2, fe_add<P>(fe_mul<P>(beta, LD(q.sigma2, i)), bg));
        p2 = fe_mul<P>(p2, fe_add<P>(fe_mul<P>(beta, LD(q.sigma3, i)), cg));
        const Fe<P> a2 = fe_sqr<P>(alpha);
        Fe<P> p3 = fe_mul<P>(fe_mul<P>(fe_sub<P>(z1, fe_one<P>()), l1), a2);
        acc = fe_add<P>(acc, fe_add<P>(fe_sub<P>(p1, p2), p3));
        // keys/lookup.rs:81-122
        const Fe<P> a3 = fe_mul<P>(a2, alpha), a4 = fe_mul<P>(a3, alpha), a5 = fe_mul<P>(a4, alpha);
        const Fe<P> opd = fe_add<P>(delta, fe_one<P>());
        const Fe<P> eopd = fe_mul<P>(eps, opd);
        const Fe<P> z2 = LD(q.z2, i), z2n = LD(q.z2, j);
        const Fe<P> t = LD(q.t, i), tn = LD(q.t, j), h1 = LD(q.h1, i), h1n = LD(q.h1, j), h2 = LD(q.h2, i);
        Fe<P> k1 = fe_mul<P>(fe_mul<P>(a3, z2), opd);
        k1 = fe_mul<P>(k1, fe_add<P>(eps, fe_mul<P>(LD(q.q_lookup, i), c)));
        k1 = fe_mul<P>(k1, fe_add<P>(fe_add<P>(eopd, t), fe_mul<P>(delta, tn)));
        Fe<P> k2 = fe_mul<P>(a3, z2n);
        k2 = fe_mul<P>(k2, fe_add<P>(fe_add<P>(eopd, h1), fe_mul<P>(delta, h2)));
        k2 = fe_mul<P>(k2, fe_add<P>(fe_add<P>(eopd, h2), fe_mul<P>(delta, h1n)));
        Fe<P> k3 = fe_mul<P>(fe_mul<P>(a4, fe_sub<P>(z2, fe_one<P>())), l1);
        Fe<P> k4 = fe_mul<P>(fe_mul<P>(a5, LD(q.q_table, i)), t);
        acc = fe_add<P>(acc, fe_add<P>(fe_add<P>(fe_sub<P>(k1, k2), k3), k4));
    }
#undef LD
    // quotient_poly.rs:220-224: times zh_coset[i]^-1; x^n - 1 takes 4 values on the 4n coset
    Fe<P> zi = arg_fe<P>(q.zh_inv[i & 3]);
    fe_store<P>((Fe<P>*)q.out + i, fe_mul<P>(acc, zi));
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
// out[i] = in[i] * z^i
template <class P>
__global__ void k_mul_pow(const Fe<P>* in, Fe<P>* out, size_t n, Fe<P> z) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * PW_E;
    if (i0 >= n) return;
    Fe<P> pw = fe_pow_u64<P>(z, i0);
#pragma unroll 1
    for (int e = 0; e < PW_E && i0 + e < n; ++e) {
        fe_store<P>(out + i0 + e, fe_mul<P>(fe_load<P>(in + i0 + e), pw));
        pw = fe_mul<P>(pw, z);
    }
}
// out[j] = S[j+1] * zinv^(j+1), j < len-1 ; zero beyond
template <class P>
__global__ void k_witness_finish(const Fe<P>* S, Fe<P>* out, size_t len, size_t cap, Fe<P> zinv) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t j0 = t * PW_E;
    if (j0 >= cap) return;
    Fe<P> pw = fe_pow_u64<P>(zinv, j0 + 1);
#pragma unroll 1
    for (int e = 0; e < PW_E && j0 + e < cap; ++e) {
        size_t j = j0 + e;
        Fe<P> v = fe_zero<P>();
        if (j + 1 < len) v = fe_mul<P>(fe_load<P>(S + j + 1), pw);
        fe_store<P>(out + j, v);
        pw = fe_mul<P>(pw, zinv);
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

template <class P> static int trim_len_t(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len) {
    ZKT_HIP(c, hipMemsetAsync(d_len, 0, 4, c->stream));
    if (!n) return ZKT_OK;
    const size_t tail = n > 1024 ? n - 1024 : 0;
    hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(n - tail)), dim3(256), 0, c->stream, (const Fe<P>*)p, tail, n, d_len, 0);
    ZKT_HIP(c, hipGetLastError());
    if (tail) {  // full scan, a no-op unless the top 1024 coefficients were all zero
        hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(tail)), dim3(256), 0, c->stream, (const Fe<P>*)p, (size_t)0, tail, d_len, 1);
        ZKT_HIP(c, hipGetLastError());
    }
    return ZKT_OK;
}
int poly_trim_len(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len) { ZKT_DISPATCH(c, trim_len_t, p, n, d_len); }

template <class P> static int add_blinders_t(zkt_ctx* c, void* p, const uint32_t* d_len, const void* bl, int k) {
    hipLaunchKernelGGL(k_add_blinders<P>, dim3(1), dim3(64), 0, c->stream, (Fe<P>*)p, d_len, (const Fe<P>*)bl, k);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_add_blinders(zkt_ctx* c, void* p, const uint32_t* d_len, const void* d_blinders, int k, size_t) {
    ZKT_DISPATCH(c, add_blinders_t, p, d_len, d_blinders, k);
}

template <class P> static int lincomb_t(zkt_ctx* c, const LinCombArgs& a, void* out, size_t n) {
    hipLaunchKernelGGL(k_lincomb<P>, dim3(nblocks(n)), dim3(256), 0, c->stream, a, (Fe<P>*)out, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_lincomb(zkt_ctx* c, const LinCombArgs& a, void* out, size_t n) { ZKT_DISPATCH(c, lincomb_t, a, out, n); }

template <class P> static int eval_many_t(zkt_ctx* c, const EvalArgs& a, void* d_partials, void* d_results) {
    uint64_t maxlen = 1;
    for (int k = 0; k < a.count; ++k) if (a.len[k] > maxlen) maxlen = a.len[k];
    int nblk = (int)((maxlen + EV_SEG - 1) / EV_SEG);
    hipLaunchKernelGGL(k_eval_partial<P>, dim3(nblk, a.count), dim3(256), 0, c->stream, a, (Fe<P>*)d_partials, nblk);
    ZKT_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_eval_final<P>, dim3(a.count), dim3(256), 0, c->stream, (const Fe<P>*)d_partials, nblk, (Fe<P>*)d_results);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int poly_eval_many(zkt_ctx* c, const EvalArgs& a, void* d_partials, void* d_results) { ZKT_DISPATCH(c, eval_many_t, a, d_partials, d_results); }

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

template <class P> static int quotient_t(zkt_ctx* c, const QuotientArgs& a) {
    ProfScope prof(c, "quotient");
    hipLaunchKernelGGL(k_quotient<P>, dim3(nblocks(a.n4)), dim3(256), 0, c->stream, a);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int quotient_pointwise(zkt_ctx* c, const QuotientArgs& a) { ZKT_DISPATCH(c, quotient_t, a); }

template <class P> static int quot_split_t(zkt_ctx* c, const void* q, size_t n, const void* b0b1, void* lo, void* mid, void* hi, uint32_t* d_status) {
    // d_status[0] = error bits ; d_status[4..7] = lens (lo, mid, hi, q)
    const Fe<P>* Q = (const Fe<P>*)q;
    const size_t cap = n + 8, chunk = n + 2;
    ZKT_HIP(c, hipMemsetAsync(lo, 0, cap * 32, c->stream));
    ZKT_HIP(c, hipMemsetAsync(mid, 0, cap * 32, c->stream));
    ZKT_HIP(c, hipMemsetAsync(hi, 0, cap * 32, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(lo, Q, chunk * 32, hipMemcpyDeviceToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(mid, Q + chunk, chunk * 32, hipMemcpyDeviceToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(hi, Q + 2 * chunk, chunk * 32, hipMemcpyDeviceToDevice, c->stream));
    uint32_t* lens = d_status + 4;
    int rc;
    if ((rc = trim_len_t<P>(c, lo, chunk, lens + 0))) return rc;
    if ((rc = trim_len_t<P>(c, mid, chunk, lens + 1))) return rc;
    if ((rc = trim_len_t<P>(c, hi, chunk, lens + 2))) return rc;
    {   // only "is anything non-zero at or above 3(n+2)?" matters for the quotient itself
        ZKT_HIP(c, hipMemsetAsync(lens + 3, 0, 4, c->stream));
        const size_t lo = 3 * chunk, hi = 4 * n;
        hipLaunchKernelGGL(k_trim_len<P>, dim3(nblocks(hi - lo)), dim3(256), 0, c->stream, (const Fe<P>*)q, lo, hi, lens + 3, 0);
        ZKT_HIP(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_quot_blind<P>, dim3(1), dim3(64), 0, c->stream, (Fe<P>*)lo, (Fe<P>*)mid, (Fe<P>*)hi, lens, (const Fe<P>*)b0b1, (uint32_t)n, d_status);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int quotient_split_blind(zkt_ctx* c, const void* q, size_t n, const void* d_b0b1, void* q_lo, void* q_mid, void* q_hi, uint32_t* d_status) {
    ZKT_DISPATCH(c, quot_split_t, q, n, d_b0b1, q_lo, q_mid, q_hi, d_status);
}

template <class P> static int open_witness_t(zkt_ctx* c, const void* p, size_t len, const uint32_t* z, const uint32_t* zinv, void* ta, void* tb, void* scan_tmp, void* out) {
    if (len == 0) return ZKT_OK;
    unsigned blocks = nblocks((len + PW_E - 1) / PW_E);
    hipLaunchKernelGGL(k_mul_pow<P>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<P>*)p, (Fe<P>*)ta, len, host_fe<P>(z));
    ZKT_HIP(c, hipGetLastError());
    int rc = scan_t<P, OpAdd>(c, (const Fe<P>*)ta, (Fe<P>*)tb, len, true, (Fe<P>*)scan_tmp);
    if (rc) return rc;
    hipLaunchKernelGGL(k_witness_finish<P>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<P>*)tb, (Fe<P>*)out, len, len, host_fe<P>(zinv));
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}
int open_witness(zkt_ctx* c, const void* p, size_t len, const uint32_t z[8], const uint32_t z_inv[8], void* d_tmp_a, void* d_tmp_b, void* d_scan_tmp, void* out) {
    ZKT_DISPATCH(c, open_witness_t, p, len, z, z_inv, d_tmp_a, d_tmp_b, d_scan_tmp, out);
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
int lookup_expand(zkt_ctx* c, const void* d_keys_insertion, const uint32_t* d_starts, uint32_t nkeys, void* out, size_t n) {
    ZKT_DISPATCH(c, lookup_expand_t, d_keys_insertion, d_starts, nkeys, out, n);
}

}  // namespace zkt
