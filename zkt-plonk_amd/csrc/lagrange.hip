// Lagrange-basis committer key: commitments of polynomials that the prover holds as EVALUATIONS.
//
// The reference commits to every polynomial through its coefficients (plonk-core/src/proof_system/prove.rs:133-135,
// 178-180,249-251: poly_from_evals -> add_blinders_to_poly -> PC::commit = one dense MSM over n + k scalars each).  Four of
// those polynomials are piecewise constant as evaluation vectors: the lookup table t (its values, then zeros:
// lookup/table.rs:52-61), the sorted halves h1 / h2 (runs of equal table values: lookup/multiset.rs:103-146) and the
// lookup grand product z2 (its ratio is 1 wherever f, t, h1, h2 stand still: lookup/mod.rs:94-154).  With
//     [L_i(tau)] G   the key in the Lagrange basis of the circuit's domain (the inverse DFT of the powers [tau^j] G),
//     S_k = sum_{i < k} [L_i(tau)] G   its prefix sums,
// Abel summation turns  commit(p) = sum_i e_i [L_i(tau)] G  into  sum_{k=1..n} (e_{k-1} - e_k) S_k  (e_n := 0): an MSM
// whose scalars are the DIFFERENCES of neighbouring evaluations -- zero inside every run, so its cost follows the number
// of runs (a few thousand), not n.  The same group element, hence the same bytes; a dense evaluation vector simply costs
// what the coefficient form costs.  Blinders b_j X^(n+j) - b_j X^j (prove.rs:472-483) ride along as k extra bases
// V_j = [tau^(n+j)] G - [tau^j] G.
//
// This file builds that second base table once per (key, domain size): the inverse DFT over G1 (radix-2 DIF, one scalar
// multiplication per butterfly: (n/2) log n of them, ~0.5 s at n = 2^20 on BN254), the prefix sums, the blinder points,
// then the window multiples exactly as for the powers (msm_table_finish).  1/n is NOT applied to the points: the scalars
// carry it (poly.hip k_lagrange_scalars).  The table holds R^-1-scaled bases like the first one (msm.hip header), which
// is free here: the transform is linear and starts from table[0].
#include "ctx.hpp"
#include "ec.hpp"
#include "hostec.hpp"
#include "msm.hpp"

#include <vector>

namespace zkt {

// A[j] = table[0][j] as an XYZZ point with arkworks-form coordinates (the table keeps canonical R' words)
template <class C>
__global__ void k_lag_init(const Affine<typename C::Fq>* table, Xyzz<typename C::Fq>* A, size_t n) {
    using Q = typename C::Fq;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Affine<Q> p = aff_load<Q>(table + j);
    if (!aff_is_inf<Q>(p)) {
        p.x = fx_to_ark<Q>(fx_unpack<Q>(p.x));
        p.y = fx_to_ark<Q>(fx_unpack<Q>(p.y));
    }
    xyzz_store<Q>(A + j, xyzz_from_affine<Q>(p));
}

// tw[j] = w^j as a canonical integer (w = omega^-1 in Montgomery form)
template <class R>
__global__ void k_lag_twiddles(Fe<R>* tw, size_t half, Fe<R> w) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    fe_store<R>(tw + j, fe_from_mont<R>(fe_pow_u64<R>(w, (uint64_t)j)));
}

template <class Q>
ZKT_HD Xyzz<Q> xyzz_neg(const Xyzz<Q>& p) {
    Xyzz<Q> r = p;
    r.y = fe_neg<Q>(p.y);
    return r;
}

// [k] P by the non-adjacent form read off 3k and k (digit i = bit i+1 of 3k minus bit i+1 of k): one doubling per bit, an
// addition every third bit on average, no table.  k canonical, below 2^(32 N - 2).
template <class Q, class R>
ZKT_D Xyzz<Q> xyzz_scalar_mul(const Xyzz<Q>& P, const Fe<R>& k) {
    constexpr int N = R::N;
    uint32_t h[N + 1];
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t t = (uint64_t)k.v[i] * 3u + carry;
        h[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    h[N] = carry;
    const Xyzz<Q> Pn = xyzz_neg<Q>(P);
    Xyzz<Q> acc = xyzz_identity<Q>();
#pragma unroll 1
    for (int li = N; li >= 0; --li) {   // (a word of h and of k per 32 doublings: the indexed reads cost nothing here)
        const uint32_t hw = h[li], kw = li < N ? k.v[li] : 0u;
        if (li == N && hw == 0u) continue;
#pragma unroll 1
        for (int b = (li == N ? 1 : 31); b >= (li == 0 ? 1 : 0); --b) {
            acc = xyzz_double<Q>(acc);
            const uint32_t hb = (hw >> b) & 1u, kb = (kw >> b) & 1u;
            if (hb != kb) acc = xyzz_add<Q>(acc, hb ? P : Pn);
        }
    }
    return acc;
}

// one level of the in-place decimation-in-frequency transform: (u, v) -> (u + v, [w^(j stride)] (u - v))
template <class C>
__global__ __launch_bounds__(128) void k_lag_level(Xyzz<typename C::Fq>* A, size_t n, size_t h, const Fe<typename C::Fr>* tw,
                                                   size_t stride) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    const size_t blk = t / h, j = t % h;
    const size_t i0 = blk * 2 * h + j, i1 = i0 + h;
    const Xyzz<Q> u = xyzz_load<Q>(A + i0), v = xyzz_load<Q>(A + i1);
    xyzz_store<Q>(A + i0, xyzz_add<Q>(u, v));
    Xyzz<Q> d = xyzz_add<Q>(u, xyzz_neg<Q>(v));
    if (j != 0 && !xyzz_is_identity<Q>(d)) d = xyzz_scalar_mul<Q, R>(d, fe_load<R>(tw + j * stride));
    xyzz_store<Q>(A + i1, d);
}

ZKT_D size_t lag_bitrev(size_t i, int bits) {
    return bits ? (size_t)(__brev((uint32_t)i) >> (32 - bits)) : 0;
}

// the transform leaves frequency i at A[bitrev(i)]; seg[s] = sum of the frequencies [s len, (s + 1) len)
template <class C>
__global__ __launch_bounds__(64) void k_lag_seg_sum(const Xyzz<typename C::Fq>* A, int log_n, size_t len, size_t nseg,
                                                    Xyzz<typename C::Fq>* seg) {
    using Q = typename C::Fq;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    Xyzz<Q> acc = xyzz_identity<Q>();
#pragma unroll 1
    for (size_t i = s * len; i < (s + 1) * len; ++i) acc = xyzz_add<Q>(acc, xyzz_load<Q>(A + lag_bitrev(i, log_n)));
    xyzz_store<Q>(seg + s, acc);
}

// out[i] = S_(i+1) = pre[s] + the frequencies of segment s up to and including i, affine
template <class C>
__global__ __launch_bounds__(64) void k_lag_prefix(const Xyzz<typename C::Fq>* A, int log_n, size_t len, size_t nseg,
                                                   const Xyzz<typename C::Fq>* pre, Affine<typename C::Fq>* out) {
    using Q = typename C::Fq;
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    Xyzz<Q> acc = xyzz_load<Q>(pre + s);
#pragma unroll 1
    for (size_t i = s * len; i < (s + 1) * len; ++i) {
        acc = xyzz_add<Q>(acc, xyzz_load<Q>(A + lag_bitrev(i, log_n)));
        aff_store<Q>(out + i, xyzz_to_affine<Q>(acc));
    }
}

// out[n + t] = table[0][n + t] - table[0][t]: the base a blinder at X^(n+t) (minus itself at X^t) multiplies
template <class C>
__global__ void k_lag_blinder_points(const Affine<typename C::Fq>* table, size_t n, size_t extra, Affine<typename C::Fq>* out) {
    using Q = typename C::Fq;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= extra) return;
    Affine<Q> hi = aff_load<Q>(table + n + t), lo = aff_load<Q>(table + t);
    if (!aff_is_inf<Q>(hi)) {
        hi.x = fx_to_ark<Q>(fx_unpack<Q>(hi.x));
        hi.y = fx_to_ark<Q>(fx_unpack<Q>(hi.y));
    }
    Xyzz<Q> acc = xyzz_from_affine<Q>(hi);
    if (!aff_is_inf<Q>(lo)) {
        lo.x = fx_to_ark<Q>(fx_unpack<Q>(lo.x));
        lo.y = fe_neg<Q>(fx_to_ark<Q>(fx_unpack<Q>(lo.y)));
        acc = xyzz_add_mixed<Q>(acc, lo);
    }
    aff_store<Q>(out + n + t, xyzz_to_affine<Q>(acc));
}

constexpr size_t LAG_MAX_SEGMENTS = 4096;   // segment totals the host scans
constexpr size_t LAG_MAX_EXTRA = 8;         // blinder bases kept (the prover uses at most three per polynomial)

template <class C>
static int lagrange_build_t(zkt_ctx* c, int log_n) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    MsmState& st = *c->msm;
    const size_t n = (size_t)1 << log_n;
    const size_t extra = std::min(st.count - n, LAG_MAX_EXTRA);
    const size_t count2 = n + extra;
    int rc;
    void *A = nullptr, *tw = nullptr, *seg = nullptr, *table2 = nullptr;
    if ((rc = dev_alloc(c, &table2, (size_t)st.W * count2 * sizeof(Affine<Q>)))) return rc;
    auto release = [&](int code) {
        dev_free(c, A);
        dev_free(c, tw);
        dev_free(c, seg);
        if (code) dev_free(c, table2);
        return code;
    };
    if ((rc = dev_alloc(c, &A, n * sizeof(Xyzz<Q>)))) return release(rc);
    hipLaunchKernelGGL(k_lag_init<C>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       (const Affine<Q>*)st.table, (Xyzz<Q>*)A, n);
    if (log_n >= 1) {
        const size_t half = n / 2;
        if ((rc = dev_alloc(c, &tw, half * sizeof(Fe<R>)))) return release(rc);
        const Fe<R> winv = fe_inv_host<R>(root_of_unity<R>(log_n));
        hipLaunchKernelGGL(k_lag_twiddles<R>, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, c->stream, (Fe<R>*)tw, half, winv);
        for (size_t h = half; h >= 1; h >>= 1)
            hipLaunchKernelGGL(k_lag_level<C>, dim3((unsigned)((half + 127) / 128)), dim3(128), 0, c->stream, (Xyzz<Q>*)A, n, h,
                               (const Fe<R>*)tw, half / h);
    }
    if (hipGetLastError() != hipSuccess) return release(set_err(c, ZKT_ERR_HIP, "Lagrange key: launch failed"));
    // prefix sums: segment totals on the device, their scan on the host (a few thousand additions), the rest on the device
    const size_t nseg = std::min(n, LAG_MAX_SEGMENTS), len = n / nseg;
    if ((rc = dev_alloc(c, &seg, 2 * nseg * sizeof(Xyzz<Q>)))) return release(rc);
    Xyzz<Q>* d_tot = (Xyzz<Q>*)seg;
    Xyzz<Q>* d_pre = d_tot + nseg;
    hipLaunchKernelGGL(k_lag_seg_sum<C>, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, c->stream, (const Xyzz<Q>*)A, log_n, len,
                       nseg, d_tot);
    std::vector<Xyzz<Q>> tot(nseg), pre(nseg);
    if (hipMemcpyAsync(tot.data(), d_tot, nseg * sizeof(Xyzz<Q>), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return release(set_err(c, ZKT_ERR_HIP, "Lagrange key: the transform failed"));
    Xyzz<Q> run = xyzz_identity<Q>();
    for (size_t s = 0; s < nseg; ++s) {
        pre[s] = run;
        run = xyzz_add<Q>(run, tot[s]);
    }
    if (hipMemcpyAsync(d_pre, pre.data(), nseg * sizeof(Xyzz<Q>), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return release(set_err(c, ZKT_ERR_HIP, "Lagrange key: upload failed"));
    hipLaunchKernelGGL(k_lag_prefix<C>, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, c->stream, (const Xyzz<Q>*)A, log_n, len,
                       nseg, (const Xyzz<Q>*)d_pre, (Affine<Q>*)table2);
    if (extra)
        hipLaunchKernelGGL(k_lag_blinder_points<C>, dim3(1), dim3(64), 0, c->stream, (const Affine<Q>*)st.table, n, extra,
                           (Affine<Q>*)table2);
    if (hipGetLastError() != hipSuccess) return release(set_err(c, ZKT_ERR_HIP, "Lagrange key: launch failed"));
    if ((rc = msm_table_finish(c, table2, count2))) return release(rc);   // synchronises: `pre` may go
    if (!st.table2_borrowed) dev_free(c, st.table2);
    st.table2_borrowed = false;
    st.table2 = table2;
    st.count2 = count2;
    st.lag_log_n = log_n;
    return release(ZKT_OK);
}

// Makes the table for the domain of size 2^log_n available if the key allows it (not sharded, at least n + 1 powers).
// Returns ZKT_OK either way; lagrange_ready tells whether evaluations can be committed directly.
int lagrange_ensure(zkt_ctx* c, int log_n) {
    if (!c->msm) return ZKT_OK;
    MsmState& st = *c->msm;
    if (st.table2 && st.lag_log_n == log_n) return ZKT_OK;
    if (st.lag_failed && st.lag_log_n == log_n) return ZKT_OK;
    const size_t n = (size_t)1 << log_n;
    const bool whole_key = !c->sharded() && st.slice_off == 0 && st.total == st.count;
    if (!whole_key || st.count <= n || log_n > 30 || (uint64_t)st.W * (n + LAG_MAX_EXTRA) >= ((uint64_t)1 << 31)) {
        if (!st.table2_borrowed) dev_free(c, st.table2);
        st.table2_borrowed = false;
        st.table2 = nullptr;
        st.count2 = 0;
        st.lag_log_n = log_n;
        st.lag_failed = true;
        return ZKT_OK;
    }
    st.lag_failed = false;
    ++c->msm_epoch;
    if (c->curve == ZKT_CURVE_BN254) return lagrange_build_t<Bn254Curve>(c, log_n);
    return lagrange_build_t<Bls381Curve>(c, log_n);
}

bool lagrange_ready(const zkt_ctx* c, int log_n) {
    return c->msm && c->msm->table2 && c->msm->lag_log_n == log_n;
}
size_t lagrange_bases(const zkt_ctx* c) { return c->msm ? c->msm->count2 : 0; }

}  // namespace zkt
