// NTT kernels + plan construction (see ntt.hpp for the decomposition).
#include "ctx.hpp"
#include "hostinv.hpp"
#include <cstdlib>

namespace zkt {

constexpr int TILE_LOG = 10;          // elements per workgroup tile
constexpr int TILE = 1 << TILE_LOG;   // 1024 x 32 B = 32 KiB of LDS
constexpr int NTT_THREADS = 256;      // 4 elements per thread
constexpr int EPT = TILE / NTT_THREADS;

// LDS tile: every element is held as nine 29-bit limbs (fx.hpp), lazily reduced (< 2p between steps),
// in three planes (limbs 0-3, 4-7, 8) so that consecutive lanes touch consecutive 16-byte / 4-byte slots.
template <class P>
ZKT_D Fx<P> tile_get(const uint4* lo, const uint4* hi, const uint32_t* top, int64_t idx) {
    static_assert(FxP<P>::L == 9, "scalar fields use nine limbs");
    const uint4 a = lo[idx], b = hi[idx];
    Fx<P> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = top[idx];
    return r;
}
template <class P>
ZKT_D void tile_put(uint4* lo, uint4* hi, uint32_t* top, int64_t idx, const Fx<P>& x) {
    lo[idx] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    hi[idx] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    top[idx] = x.l[8];
}

ZKT_D uint32_t bitrev32(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// inner twiddles of a pass in LDS: R/2 entries.  MODE 2 (Shoup): each the pair (w, wq = floor(w 2^261 / p)) of
// fx_mul_shoup as eighteen 29-bit limbs in the tile's three-plane layout (72 bytes per entry, no Montgomery factor).
// MODE 1 / 0: the Montgomery form w 2^261 mod p of the earlier rounds (product = fx_mul_inl), as nine limbs (36 bytes) or,
// MODE 0, as eight packed words unpacked at every use (32 bytes: at R = 256 that keeps a fourth workgroup per CU).
constexpr int NTT_W_PACKED = 0, NTT_W_LIMBS = 1, NTT_W_SHOUP = 2;
template <class P, int MODE>
struct WTile {
    const uint4* lo;
    const uint4* hi;
    const uint32_t* top;
    const uint4* qlo;      // the quotient words stay in global memory (three planes of R/2 entries, a few KiB that live in
    const uint4* qhi;      // the vector L1): with them in LDS as well the tile would cost the fourth workgroup per CU
    const uint32_t* qtop;
    // d: the butterfly's difference (limbs <= 2^31.33, value < 16p); returns d * w, normalised, < 3p
    ZKT_D Fx<P> mul(int e, const Fx<P>& d) const {
        if constexpr (MODE == NTT_W_SHOUP) {
            return fx_mul_shoup<P>(d, tile_get<P>(lo, hi, top, e), tile_get<P>(qlo, qhi, qtop, e));
        } else if constexpr (MODE == NTT_W_LIMBS) {
            return fx_mul_inl<P>(tile_get<P>(lo, hi, top, e), d);
        } else {
            Fe<P> w;
            const uint4 a = lo[e], b = hi[e];
            w.v[0] = a.x; w.v[1] = a.y; w.v[2] = a.z; w.v[3] = a.w;
            w.v[4] = b.x; w.v[5] = b.y; w.v[6] = b.z; w.v[7] = b.w;
            return fx_mul_inl<P>(fx_unpack<P>(w), d);
        }
    }
};

// One radix-2^G DIF step on 2^G elements held in registers (inputs < 3p, limbs normalised).
//   level L: block size m = R >> L; element i sits at row blk*m + i*sub + off, sub = m >> G.
//   sub-level l pairs (i, i + h), h = 2^(G-1-l); twiddle W_R^(((i & (h-1))*sub + off) << (L + l)).
// Nothing propagates a carry inside a group: sums are limb-wise (limbs < 2^30 after sub-level 0, < 2^31 after 1), a
// difference that feeds a twiddle product is u + K p - v with K p spread so that no limb goes negative (fx_sub_lazy: the
// subtrahend normalised, K = 4 >= 3 + 1; fx_sub_lazy_wide<8, 30>: a sub-level-0 result, < 7p with limbs < 2^30), and the
// product brings it back below 3p with normalised limbs.  In the LAST step of a pass (sub == 1, off == 0) the twiddle
// index depends on i only and W^0 products are skipped at compile time: those differences take a carry pass.
template <class P, int G, bool LASTSTEP, class W>
ZKT_D void dif_group(Fx<P>* x, const W& w, int off, int sub, int L) {
    static_assert(G == 1 || G == 2, "radix 2 or 4 register groups");
#pragma unroll
    for (int l = 0; l < G; ++l) {
        const int h = 1 << (G - 1 - l);
#pragma unroll
        for (int i = 0; i < (1 << G); ++i) {
            if ((i & h) == 0) {
                const Fx<P> u = x[i], v = x[i + h];
                x[i] = fx_add_lazy<P>(u, v);
                if (LASTSTEP && (i & (h - 1)) == 0) {   // twiddle is W^0 = 1
                    if (l == 0) x[i + h] = fx_sub<P, 4>(u, v);          // < 7p
                    else x[i + h] = fx_sub<P, 8>(u, v);                  // < 15p
                } else {
                    Fx<P> d;
                    if (l == 0) d = fx_sub_lazy<P, 4>(u, v);             // < 7p, limbs < 3 * 2^29
                    else d = fx_sub_lazy_wide<P, 8, 30>(u, v);           // < 15p, limbs < 2^31.33
                    const int e = ((i & (h - 1)) * sub + off) << (L + l);
                    x[i + h] = w.mul(e, d);
                }
            }
        }
    }
}

template <class P, int LOG_R, int G, int L, class W>
ZKT_D void dif_step(uint4* lo, uint4* hi, uint32_t* top, const W& w_inner, int tid) {
    constexpr int R = 1 << LOG_R;
    constexpr int T = TILE >> LOG_R;
    constexpr int m = R >> L;
    constexpr int sub = m >> G;
    constexpr int UNITS = EPT >> G;  // independent radix-2^G groups per thread
    constexpr bool LASTSTEP = (L + G == LOG_R);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        int q = tid * UNITS + u;
        int c = q % T;
        int rest = q / T;
        int off = rest % sub;
        int blk = rest / sub;
        int row0 = blk * m + off;
        Fx<P> x[1 << G];
#pragma unroll
        for (int i = 0; i < (1 << G); ++i) x[i] = tile_get<P>(lo, hi, top, (row0 + i * sub) * T + c);
        dif_group<P, G, LASTSTEP>(x, w_inner, off, sub, L);
        // values that did not end on a twiddle product are lazy sums (limbs < 2^31, < 16p): back below 3p, normalised
#pragma unroll
        for (int i = 0; i < (1 << G); ++i) {
            const bool grown = LASTSTEP || ((i & 1) == 0);
            tile_put<P>(lo, hi, top, (row0 + i * sub) * T + c, grown ? fx_reduce_lazy<P>(x[i]) : x[i]);
        }
    }
}

template <class P, int LOG_R, class W>
ZKT_D void dif_all(uint4* lo, uint4* hi, uint32_t* top, const W& w, int tid) {
    if constexpr (LOG_R == 5) {
        dif_step<P, 5, 2, 0>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 5, 2, 2>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 5, 1, 4>(lo, hi, top, w, tid);
    } else if constexpr (LOG_R == 6) {
        dif_step<P, 6, 2, 0>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 6, 2, 2>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 6, 2, 4>(lo, hi, top, w, tid);
    } else if constexpr (LOG_R == 7) {
        dif_step<P, 7, 2, 0>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 7, 2, 2>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 7, 2, 4>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 7, 1, 6>(lo, hi, top, w, tid);
    } else if constexpr (LOG_R == 8) {
        dif_step<P, 8, 2, 0>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 8, 2, 2>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 8, 2, 4>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 8, 2, 6>(lo, hi, top, w, tid);
    } else {
        static_assert(LOG_R == 9, "unsupported radix");
        dif_step<P, 9, 2, 0>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 9, 2, 2>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 9, 2, 4>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 9, 2, 6>(lo, hi, top, w, tid); __syncthreads();
        dif_step<P, 9, 1, 8>(lo, hi, top, w, tid);
    }
}

// One pass: R-point transforms of a [R][T] tile.  LAST selects the transposing pass.
// Tables (w_inner, in_row, tw, out_row) are canonical packed words in R' = 2^261 Montgomery form, the data
// stay in arkworks' R = 2^256 form: data * table / R' keeps the data's form.
template <class P, int LOG_R, bool LAST, int MODE>
__global__ __launch_bounds__(NTT_THREADS, (LOG_R <= 7 ? 4 : 3)) void k_ntt_pass(NttPassArgs a) {   // workgroups per CU the LDS allows
    constexpr int R = 1 << LOG_R;
    constexpr int LOG_T = TILE_LOG - LOG_R;
    constexpr int T = 1 << LOG_T;
    __shared__ uint4 lds_lo[TILE];
    __shared__ uint4 lds_hi[TILE];
    __shared__ uint32_t lds_top[TILE];
    typedef WTile<P, MODE> WT;
    constexpr bool SHOUP = MODE == NTT_W_SHOUP;
    __shared__ uint4 lds_w_lo[R / 2];
    __shared__ uint4 lds_w_hi[R / 2];
    __shared__ uint32_t lds_w_top[MODE != NTT_W_PACKED ? R / 2 : 1];

    const int tid = threadIdx.x;
    // blockIdx.y = polynomial of the batch (uniform: scalar selects, no indexed access to the kernel arguments)
    const uint32_t py = blockIdx.y;
    const void* in_p = py == 0 ? a.in[0] : py == 1 ? a.in[1] : py == 2 ? a.in[2] : a.in[3];
    void* out_p = py == 0 ? a.out[0] : py == 1 ? a.out[1] : py == 2 ? a.out[2] : a.out[3];
    const uint64_t in_len = py == 0 ? a.in_len[0] : py == 1 ? a.in_len[1] : py == 2 ? a.in_len[2] : a.in_len[3];
    if (a.in_raw) in_p = reinterpret_cast<const char*>(a.in[0]) + (uint64_t)py * 36u * a.raw_n;
    if (a.out_raw) out_p = reinterpret_cast<char*>(a.out[0]) + (uint64_t)py * 36u * a.raw_n;
    const Fe<P>* in = reinterpret_cast<const Fe<P>*>(in_p);
    Fe<P>* out = reinterpret_cast<Fe<P>*>(out_p);
    const Fe<P>* w_inner = reinterpret_cast<const Fe<P>*>(a.w_inner);
    const Fe<P>* in_row = reinterpret_cast<const Fe<P>*>(a.in_row);
    const Fe<P>* tw = reinterpret_cast<const Fe<P>*>(a.tw);
    const Fe<P>* out_row = reinterpret_cast<const Fe<P>*>(a.out_row);

    for (int i = tid; i < R / 2; i += NTT_THREADS) {
        if constexpr (SHOUP) {   // six planes of R/2 entries: w (lo, hi, top), then wq
            const uint4* wl = reinterpret_cast<const uint4*>(a.w_inner_s);
            tile_put<P>(lds_w_lo, lds_w_hi, lds_w_top, i, tile_get<P>(wl, wl + R / 2, reinterpret_cast<const uint32_t*>(wl + R), i));
        } else if constexpr (MODE == NTT_W_LIMBS) {
            tile_put<P>(lds_w_lo, lds_w_hi, lds_w_top, i, fx_unpack<P>(fe_load<P>(w_inner + i)));
        } else {
            const Fe<P> w = fe_load<P>(w_inner + i);
            lds_w_lo[i] = make_uint4(w.v[0], w.v[1], w.v[2], w.v[3]);
            lds_w_hi[i] = make_uint4(w.v[4], w.v[5], w.v[6], w.v[7]);
        }
    }
    // the table: uint4 w_lo[R/2] | uint4 w_hi[R/2] | u32 w_top[R/2] | (16-byte aligned) uint4 q_lo[R/2] | uint4 q_hi[R/2] | u32 q_top[R/2]
    const uint4* gq = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w_inner_s) + (R / 2) * 36 + ((R / 2) * 36 % 16 ? 16 - (R / 2) * 36 % 16 : 0));
    const WT lds_w{lds_w_lo, lds_w_hi, lds_w_top, gq, gq + R / 2, reinterpret_cast<const uint32_t*>(gq + R)};

    const uint64_t tile = blockIdx.x;
    uint64_t in_base, out_base, tw_base;
    uint64_t ld_r, ld_c, st_k;
    if constexpr (!LAST) {
        // position = hi*(R*S) + r*S + lo ; tile = T consecutive lo
        const uint32_t log_s = a.log_s;
        const uint64_t tiles_per_hi = (uint64_t)1 << (log_s - LOG_T);
        const uint64_t hi = tile >> (log_s - LOG_T);
        const uint64_t lo0 = (tile & (tiles_per_hi - 1)) << LOG_T;
        in_base = (hi << (LOG_R + log_s)) + lo0;
        out_base = in_base;
        ld_r = (uint64_t)1 << log_s;
        ld_c = 1;
        st_k = ld_r;
        tw_base = hi << LOG_R;  // row-shared: tw[hi*R + r]
    } else {
        // input [k1][mid][r] ; tile = T consecutive k1 for one mid ; output k1 + R1*mid + (N/R)*k
        const uint32_t log_r1 = a.log_r1, log_mid = a.log_mid;
        const uint64_t k1_tiles = (uint64_t)1 << (log_r1 - LOG_T);
        const uint64_t mid = tile >> (log_r1 - LOG_T);
        const uint64_t k10 = (tile & (k1_tiles - 1)) << LOG_T;
        in_base = (k10 << (log_mid + LOG_R)) + (mid << LOG_R);
        ld_r = 1;
        ld_c = (uint64_t)1 << (log_mid + LOG_R);
        out_base = k10 + (mid << log_r1);
        st_k = (uint64_t)1 << (a.log_n - LOG_R);
        tw_base = (k10 + (mid << log_r1)) << LOG_R;  // tile-shaped: tw[(K0 + c)*R + r]
    }

    // ---- load (coalesced along the contiguous axis), fused input scaling ----
#pragma unroll 2
    for (int e = 0; e < EPT; ++e) {
        int flat = e * NTT_THREADS + tid;
        int r, c;
        if constexpr (!LAST) {
            c = flat & (T - 1);
            r = flat >> LOG_T;
        } else {
            r = flat & (R - 1);
            c = flat >> LOG_R;
        }
        uint64_t g = in_base + (uint64_t)r * ld_r + (uint64_t)c * ld_c;
        Fx<P> x;
        if (a.in_raw) {
            const uint4* rlo = reinterpret_cast<const uint4*>(in_p);
            x = tile_get<P>(rlo, rlo + a.raw_n, reinterpret_cast<const uint32_t*>(rlo + 2 * a.raw_n), (int64_t)g);
        } else {
            x = (g < in_len) ? fx_unpack<P>(fe_load<P>(in + g)) : fx_zero<P>();
        }
        // the prover's coset transforms are fed n + 8 coefficients on a domain of 4n: three rows in four are padding,
        // and whole wavefronts see nothing but padding (a wave covers consecutive rows), so the branch is uniform
        if (in_row && (a.in_raw || g < in_len)) x = fx_mul<P>(x, fx_unpack<P>(fe_load<P>(in_row + r)));
        if (tw) {
            uint64_t ti = LAST ? (tw_base + ((uint64_t)c << LOG_R) + r) : (tw_base + r);
            x = fx_mul<P>(x, fx_unpack<P>(fe_load<P>(tw + ti)));
        }
        tile_put<P>(lds_lo, lds_hi, lds_top, r * T + c, x);
    }
    __syncthreads();

    dif_all<P, LOG_R>(lds_lo, lds_hi, lds_top, lds_w, tid);
    __syncthreads();

    // ---- store: row rho of the tile holds frequency bitrev(rho); canonicalise on the way out ----
#pragma unroll 2
    for (int e = 0; e < EPT; ++e) {
        int flat = e * NTT_THREADS + tid;
        int c = flat & (T - 1);
        int k = flat >> LOG_T;
        int rho = (int)bitrev32((uint32_t)k, LOG_R);
        Fx<P> x = tile_get<P>(lds_lo, lds_hi, lds_top, rho * T + c);
        if (out_row) x = fx_mul<P>(x, fx_unpack<P>(fe_load<P>(out_row + k)));   // < 2p
        else if (!a.out_raw) x = fx_cond_sub_p<P>(x);                            // the tile holds values < 3p
        const uint64_t at = out_base + (uint64_t)k * st_k + c;
        if (a.out_raw) {
            uint4* rlo = reinterpret_cast<uint4*>(out_p);
            tile_put<P>(rlo, rlo + a.raw_n, reinterpret_cast<uint32_t*>(rlo + 2 * a.raw_n), (int64_t)at, x);
        } else {
            fe_store<P>(out + at, fx_pack<P>(fx_cond_sub_p<P>(x)));
        }
    }
}

// table entries: arkworks R form -> canonical R' form, in place (pass-kernel tables only)
template <class P>
__global__ void k_table_to_fx(Fe<P>* t, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<P>(t + i, fx_pack<P>(fx_cond_sub_p<P>(fx_from_ark<P>(fe_load<P>(t + i)))));
}

// inner twiddles, arkworks R form -> the pairs of fx_mul_shoup: w as a plain canonical integer and
// wq = floor(w 2^261 / p) = (w 2^261 - w^) / p with w^ = w 2^261 mod p, an exact division: wq = w^ * (-p^-1) mod 2^261
template <class P>
__global__ void k_table_to_shoup(const Fe<P>* t, void* out, uint64_t n, Fx<P> npinv) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe<P> x = fe_load<P>(t + i);
    const Fx<P> w = fx_unpack<P>(fe_from_mont<P>(x));
    const Fx<P> what = fx_cond_sub_p<P>(fx_from_ark<P>(x));
    const Fx<P> wq = fx_mul_low<P>(what, npinv);
    // two groups of three planes (the LDS tile's layout), the second 16-byte aligned
    const uint64_t plane = n * 36, second = (plane + 15) & ~(uint64_t)15;
    uint4* wl = reinterpret_cast<uint4*>(out);
    tile_put<P>(wl, wl + n, reinterpret_cast<uint32_t*>(wl + 2 * n), (int64_t)i, w);
    uint4* ql = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + second);
    tile_put<P>(ql, ql + n, reinterpret_cast<uint32_t*>(ql + 2 * n), (int64_t)i, wq);
}

// Whole transform in one workgroup for n <= 1024 (plumbing sizes; not a performance path).
template <class P>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_small(const Fe<P>* in, uint64_t in_len, Fe<P>* out, int log_n,
                                                           const Fe<P>* w, const Fe<P>* in_scale,
                                                           const Fe<P>* out_scale) {
    __shared__ Fe<P> buf[TILE];
    const int n = 1 << log_n;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += NTT_THREADS) {
        Fe<P> x = ((uint64_t)i < in_len) ? fe_load<P>(in + i) : fe_zero<P>();
        if (in_scale) x = fe_mul<P>(x, fe_load<P>(in_scale + i));
        buf[i] = x;
    }
    __syncthreads();
    for (int s = 0; s < log_n; ++s) {
        const int m = n >> s, half = m >> 1;
        for (int idx = tid; idx < n / 2; idx += NTT_THREADS) {
            int blk = idx / half, j = idx % half;
            int i0 = blk * m + j, i1 = i0 + half;
            Fe<P> u = buf[i0], v = buf[i1];
            buf[i0] = fe_add<P>(u, v);
            buf[i1] = fe_mul<P>(fe_sub<P>(u, v), fe_load<P>(w + ((size_t)j << s)));
        }
        __syncthreads();
    }
    for (int k = tid; k < n; k += NTT_THREADS) {
        Fe<P> x = buf[log_n ? bitrev32((uint32_t)k, log_n) : 0];
        if (out_scale) x = fe_mul<P>(x, fe_load<P>(out_scale + k));
        fe_store<P>(out + k, x);
    }
}

// ---- table generation (once per plan) ------------------------------------------------------
// out[i] = scale * base^i
template <class P>
__global__ void k_gen_pow(Fe<P>* out, uint64_t n, Fe<P> base, Fe<P> scale) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<P>(out + i, fe_mul<P>(scale, fe_pow_u64<P>(base, i)));
}
// out[K*cols + m] = scale * w^(m*K) * row_base^K * col_base^m
template <class P>
__global__ void k_gen_tw2d(Fe<P>* out, uint64_t rows, uint32_t log_cols, Fe<P> w, Fe<P> row_base, Fe<P> col_base,
                           Fe<P> scale, int use_row, int use_col) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (rows << log_cols)) return;
    uint64_t K = i >> log_cols, m = i & (((uint64_t)1 << log_cols) - 1);
    Fe<P> x = fe_mul<P>(scale, fe_pow_u64<P>(w, m * K));
    if (use_row) x = fe_mul<P>(x, fe_pow_u64<P>(row_base, K));
    if (use_col) x = fe_mul<P>(x, fe_pow_u64<P>(col_base, m));
    fe_store<P>(out + i, x);
}

// ---- host side ---------------------------------------------------------------------------------
template <class P>
static Fe<P> host_pow_limbs(const Fe<P>& a, const uint32_t* e, int nlimbs) {
    Fe<P> r = fe_one<P>();
    bool started = false;
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        if (started) r = fe_sqr<P>(r);
        if ((e[i / 32] >> (i % 32)) & 1u) {
            r = fe_mul<P>(r, a);
            started = true;
        }
    }
    return r;
}

// FftField::get_root_of_unity(2^log_n) (ark-ff 0.3), Montgomery form
template <class P>
Fe<P> root_of_unity(int log_n) {
    uint32_t e[P::N];
    for (int i = 0; i < P::N; ++i) e[i] = P::mod(i);
    e[0] -= 1;  // p - 1 (p odd)
    for (int k = 0; k < P::TWO_ADICITY; ++k)
        for (int i = 0; i < P::N; ++i) e[i] = (e[i] >> 1) | (i + 1 < P::N ? (e[i + 1] << 31) : 0u);
    Fe<P> w = host_pow_limbs<P>(fe_from_u32<P>(P::GENERATOR), e, P::N);
    for (int k = 0; k < P::TWO_ADICITY - log_n; ++k) w = fe_sqr<P>(w);
    return w;
}
template Fe<Bn254Fr> root_of_unity<Bn254Fr>(int);
template Fe<Bls381Fr> root_of_unity<Bls381Fr>(int);

static void split_log_n(int log_n, int* npass, int* lr) {
    if (log_n <= TILE_LOG) {
        *npass = 0;
        return;
    }
    int p = log_n <= 16 ? 2 : 3;
    *npass = p;
    int base = log_n / p, rem = log_n % p;
    for (int i = 0; i < p; ++i) lr[i] = base + (i < rem ? 1 : 0);
}

template <class P>
struct PlanHolder {
    NttPlan<P> plan;
    zkt_ctx* ctx;
    ~PlanHolder() {}
};

template <class P>
static int alloc_table(zkt_ctx* c, NttPlan<P>& pl, void** p, size_t count) {
    int rc = dev_alloc(c, p, count * sizeof(Fe<P>));
    if (rc) return rc;
    pl.table_bytes += count * sizeof(Fe<P>);
    return 0;
}

template <class P>
static int gen_pow(zkt_ctx* c, void* out, uint64_t n, const Fe<P>& base, const Fe<P>& scale) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_gen_pow<P>, dim3(blocks), dim3(256), 0, c->stream, (Fe<P>*)out, n, base, scale);
    ZKT_HIP(c, hipGetLastError());
    return 0;
}
template <class P>
static int gen_tw2d(zkt_ctx* c, void* out, uint64_t rows, uint32_t log_cols, const Fe<P>& w, const Fe<P>* row_base,
                    const Fe<P>* col_base, const Fe<P>& scale) {
    uint64_t total = rows << log_cols;
    unsigned blocks = (unsigned)((total + 255) / 256);
    Fe<P> one = fe_one<P>();
    hipLaunchKernelGGL(k_gen_tw2d<P>, dim3(blocks), dim3(256), 0, c->stream, (Fe<P>*)out, rows, log_cols, w,
                       row_base ? *row_base : one, col_base ? *col_base : one, scale, row_base ? 1 : 0,
                       col_base ? 1 : 0);
    ZKT_HIP(c, hipGetLastError());
    return 0;
}

template <class P>
static int table_to_fx(zkt_ctx* c, void* t, uint64_t n) {
    if (!t || !n) return ZKT_OK;
    hipLaunchKernelGGL(k_table_to_fx<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (Fe<P>*)t, n);
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}

// Coset codes (the third key of a plan): 0 = none; 1 = the multiplicative generator g (ark-poly's coset_fft);
// ntt_class_code(log_big, cls) = g * w_{2^log_big}^cls: the points of the 2^log_big coset whose index is cls modulo
// 2^(log_big - log_n) -- the class of the big transform's outputs one GPU of a sharded proof owns.
int ntt_class_code(int log_big, int cls) { return 2 + cls + 256 * log_big; }

template <class P>
static Fe<P> coset_shift(int code) {
    Fe<P> g = fe_from_u32<P>(P::GENERATOR);
    if (code <= 1) return g;
    const int cls = (code - 2) & 255, log_big = (code - 2) >> 8;
    return fe_mul<P>(g, fe_pow_u64<P>(root_of_unity<P>(log_big), (uint64_t)cls));
}

template <class P>
static int build_plan(zkt_ctx* c, int log_n, int inverse, int coset, NttPlan<P>& pl) {
    pl.log_n = log_n;
    pl.inverse = inverse;
    pl.coset = coset;
    split_log_n(log_n, &pl.npass, pl.log_r);
    const uint64_t N = (uint64_t)1 << log_n;
    Fe<P> w = root_of_unity<P>(log_n);
    if (inverse) w = fe_inv_host<P>(w);
    Fe<P> g = coset_shift<P>(coset);
    Fe<P> ginv = fe_inv_host<P>(g);
    Fe<P> one = fe_one<P>();
    Fe<P> ninv = one;
    if (inverse) {
        Fe<P> nn = fe_zero<P>();
        nn.v[0] = (uint32_t)(N & 0xffffffffu);
        nn.v[1] = (uint32_t)(N >> 32);
        ninv = fe_inv_host<P>(fe_to_mont<P>(nn));
    }
    int rc;
    if (pl.npass == 0) {
        if (log_n >= 1) {
            if ((rc = alloc_table(c, pl, &pl.small_w, N / 2))) return rc;
            if ((rc = gen_pow<P>(c, pl.small_w, N / 2, w, one))) return rc;
        }
        if (!inverse && coset) {
            if ((rc = alloc_table(c, pl, &pl.small_in, N))) return rc;
            if ((rc = gen_pow<P>(c, pl.small_in, N, g, one))) return rc;
        }
        if (inverse) {
            if ((rc = alloc_table(c, pl, &pl.small_out, N))) return rc;
            if ((rc = gen_pow<P>(c, pl.small_out, N, coset ? ginv : one, ninv))) return rc;
        }
        return 0;
    }
    const int p = pl.npass;
    // inner twiddles W_R = w^(N/R)
    for (int i = 0; i < p; ++i) {
        int lr = pl.log_r[i];
        Fe<P> wr = fe_pow_u64<P>(w, N >> lr);
        if ((rc = alloc_table(c, pl, &pl.w_inner[i], (size_t)1 << (lr - 1)))) return rc;
        if ((rc = gen_pow<P>(c, pl.w_inner[i], (uint64_t)1 << (lr - 1), wr, one))) return rc;
    }
    // strides of the input digits: S_i = N / (R_1..R_i)
    uint64_t S[3];
    {
        int acc = 0;
        for (int i = 0; i < p; ++i) {
            acc += pl.log_r[i];
            S[i] = N >> acc;
        }
    }
    if (!inverse && coset) {
        // pass-1 input row scale g^(n1*S1)
        Fe<P> gs = fe_pow_u64<P>(g, S[0]);
        if ((rc = alloc_table(c, pl, &pl.in_row, (size_t)1 << pl.log_r[0]))) return rc;
        if ((rc = gen_pow<P>(c, pl.in_row, (uint64_t)1 << pl.log_r[0], gs, one))) return rc;
    }
    // boundary tables tw[i] (consumed by pass i): rows K_i < P_i, cols n_{i+1} < R_{i+1};
    //   entry = w_{P_{i+1}}^(n*K) * [forward coset: g^(n*S_{i+1})] * [last, inverse: N^-1 * (coset: g^-K)]
    int acc = 0;
    for (int i = 1; i < p; ++i) {
        acc += pl.log_r[i - 1];
        uint64_t rows = (uint64_t)1 << acc;           // P_i
        uint32_t log_cols = pl.log_r[i];              // R_{i+1}
        Fe<P> wp = fe_pow_u64<P>(w, N >> (acc + log_cols));  // w_{P_{i+1}}
        bool last = (i == p - 1);
        Fe<P> col_base, row_base, scale = one;
        const Fe<P>* colp = nullptr;
        const Fe<P>* rowp = nullptr;
        if (!inverse && coset) {
            col_base = fe_pow_u64<P>(g, S[i]);
            colp = &col_base;
        }
        if (inverse && last) {
            scale = ninv;
            if (coset) {
                row_base = ginv;
                rowp = &row_base;
            }
        }
        if ((rc = alloc_table(c, pl, &pl.tw[i], (size_t)(rows << log_cols)))) return rc;
        if ((rc = gen_tw2d<P>(c, pl.tw[i], rows, log_cols, wp, rowp, colp, scale))) return rc;
    }
    if (inverse && coset) {
        // last-pass output row scale g^-(P_{p-1} * k_p)
        uint64_t Pm = N >> pl.log_r[p - 1];
        Fe<P> gp = fe_pow_u64<P>(ginv, Pm);
        if ((rc = alloc_table(c, pl, &pl.out_row, (size_t)1 << pl.log_r[p - 1]))) return rc;
        if ((rc = gen_pow<P>(c, pl.out_row, (uint64_t)1 << pl.log_r[p - 1], gp, one))) return rc;
    }
    // butterfly twiddles as Shoup pairs (from the arkworks form, before it is converted below)
    {
        const Fx<P> npinv = fx_neg_p_inverse<P>();
        for (int i = 0; i < p; ++i) {
            const uint64_t cnt = (uint64_t)1 << (pl.log_r[i] - 1);
            if ((rc = dev_alloc(c, &pl.w_inner_s[i], cnt * 72 + 16))) return rc;
            pl.table_bytes += cnt * 72 + 16;
            hipLaunchKernelGGL(k_table_to_shoup<P>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream,
                               (const Fe<P>*)pl.w_inner[i], pl.w_inner_s[i], cnt, npinv);
            ZKT_HIP(c, hipGetLastError());
        }
    }
    // the pass kernels multiply lazily reduced 29-bit-limb data by these tables: keep them in R' form
    for (int i = 0; i < p; ++i)
        if ((rc = table_to_fx<P>(c, pl.w_inner[i], (uint64_t)1 << (pl.log_r[i] - 1)))) return rc;
    if ((rc = table_to_fx<P>(c, pl.in_row, (uint64_t)1 << pl.log_r[0]))) return rc;
    {
        int acc2 = 0;
        for (int i = 1; i < p; ++i) {
            acc2 += pl.log_r[i - 1];
            if ((rc = table_to_fx<P>(c, pl.tw[i], (uint64_t)1 << (acc2 + pl.log_r[i])))) return rc;
        }
    }
    if ((rc = table_to_fx<P>(c, pl.out_row, (uint64_t)1 << pl.log_r[p - 1]))) return rc;
    return 0;
}

// Which form the butterfly twiddles of a radix take (WTile).  Shoup pairs save 28 of 171 multiply-adds and the m-chain
// per product; at R >= 256 their 72 bytes per entry cost the fourth resident workgroup per CU (46 KiB of LDS instead of
// 40), which the A/B build can trade back (ZKT_NTT_MONT_FROM = first log2 radix that keeps the Montgomery form).
static int ntt_mode_for(int log_r) {
    int mont_from = 10;   // every radix uses Shoup pairs
    if (const char* e = exp_env("ZKT_NTT_MONT_FROM")) mont_from = atoi(e);
    if (log_r < mont_from) return NTT_W_SHOUP;
    return log_r <= 7 ? NTT_W_LIMBS : NTT_W_PACKED;
}

template <class P, bool LAST, int LOG_R>
static void launch_pass_r(zkt_ctx* c, const dim3& grid, const NttPassArgs& a) {
    const int mode = ntt_mode_for(LOG_R);
    if (mode == NTT_W_SHOUP) {
        hipLaunchKernelGGL((k_ntt_pass<P, LOG_R, LAST, NTT_W_SHOUP>), grid, dim3(NTT_THREADS), 0, c->stream, a);
        return;
    }
#if defined(ZKT_EXPERIMENTS)
    hipLaunchKernelGGL((k_ntt_pass<P, LOG_R, LAST, (LOG_R <= 7 ? NTT_W_LIMBS : NTT_W_PACKED)>), grid, dim3(NTT_THREADS), 0, c->stream, a);
#else
    hipLaunchKernelGGL((k_ntt_pass<P, LOG_R, LAST, NTT_W_SHOUP>), grid, dim3(NTT_THREADS), 0, c->stream, a);
#endif
}

template <class P, bool LAST>
static void launch_pass(zkt_ctx* c, int log_r, unsigned blocks, unsigned nb, const NttPassArgs& a) {
    const dim3 grid(blocks, nb);
    switch (log_r) {
        case 5: launch_pass_r<P, LAST, 5>(c, grid, a); break;
        case 6: launch_pass_r<P, LAST, 6>(c, grid, a); break;
        case 7: launch_pass_r<P, LAST, 7>(c, grid, a); break;
        case 8: launch_pass_r<P, LAST, 8>(c, grid, a); break;
        default: launch_pass_r<P, LAST, 9>(c, grid, a); break;
    }
}

// nb <= NTT_MAX_BATCH transforms of the same plan, one launch per pass (gridDim.y = polynomial): the prover's rounds
// transform two or three polynomials at a time (prove.rs:120-122,166-167; quotient_poly.rs:52-96), and a 2^20
// transform alone is a single wave of workgroups whose ramp and tail are a third of its time.
template <class P>
static int ntt_run_batch_t(zkt_ctx* c, int log_n, int inverse, int coset, int nb, const void* const* d_in, const size_t* in_len,
                           void* const* d_out) {
    if (nb < 1 || nb > NTT_MAX_BATCH) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "transform batch out of range");
    if (log_n < 0 || log_n > P::TWO_ADICITY)
        return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE,
                       "InvalidEvalDomainSize: log_size_of_group " + std::to_string(log_n) + " exceeds adicity " +
                           std::to_string(P::TWO_ADICITY));
    if (log_n > 27) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "domains above 2^27 are not supported");
    const uint64_t N = (uint64_t)1 << log_n;
    for (int y = 0; y < nb; ++y)
        if (in_len[y] > N) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "more coefficients than the domain size");
    auto key = std::make_tuple(log_n, inverse ? 1 : 0, coset);
    auto it = c->ntt_plans.find(key);
    if (it == c->ntt_plans.end()) {
        auto holder = std::make_shared<NttPlan<P>>();
        int rc = build_plan<P>(c, log_n, inverse ? 1 : 0, coset, *holder);
        if (rc) return rc;
        it = c->ntt_plans.emplace(key, std::static_pointer_cast<void>(holder)).first;
    }
    const NttPlan<P>& pl = *static_cast<const NttPlan<P>*>(it->second.get());
    const std::string prof_name = "ntt_" + std::to_string(log_n);  // e.g. "ntt_20", "ntt_22"
    ProfScope prof_all(c, prof_name.c_str(), nullptr, nb);          // counted per transform
    if (pl.npass == 0) {
        for (int y = 0; y < nb; ++y)
            hipLaunchKernelGGL(k_ntt_small<P>, dim3(1), dim3(NTT_THREADS), 0, c->stream, (const Fe<P>*)d_in[y],
                               (uint64_t)in_len[y], (Fe<P>*)d_out[y], log_n, (const Fe<P>*)pl.small_w,
                               (const Fe<P>*)pl.small_in, (const Fe<P>*)pl.small_out);
        ZKT_HIP(c, hipGetLastError());
        return 0;
    }
    int rc = ensure_buffer(c, &c->ntt_scratch, &c->ntt_scratch_bytes, N * 36 * (size_t)nb);   // raw limbs between passes
    if (rc) return rc;
    const int p = pl.npass;
    const unsigned blocks = (unsigned)(N >> TILE_LOG);
    int acc = 0;
    for (int i = 0; i < p; ++i) {
        NttPassArgs a{};
        const bool last = (i == p - 1);
        for (int y = 0; y < NTT_MAX_BATCH; ++y) {
            const int yy = y < nb ? y : 0;
            a.in[y] = (i == 0) ? d_in[yy] : c->ntt_scratch;
            a.out[y] = last ? d_out[yy] : c->ntt_scratch;
            a.in_len[y] = (i == 0) ? (uint64_t)in_len[yy] : N;
        }
        a.w_inner = pl.w_inner[i];
        a.w_inner_s = pl.w_inner_s[i];
        a.in_row = (i == 0) ? pl.in_row : nullptr;
        a.tw = pl.tw[i];
        a.out_row = last ? pl.out_row : nullptr;
        a.log_n = (uint32_t)log_n;
        a.in_raw = (i == 0) ? 0u : 1u;
        a.out_raw = last ? 0u : 1u;
        a.raw_n = N;
        acc += pl.log_r[i];
        if (!last) {
            a.log_s = (uint32_t)(log_n - acc);
            launch_pass<P, false>(c, pl.log_r[i], blocks, (unsigned)nb, a);
        } else {
            a.log_r1 = (uint32_t)pl.log_r[0];
            a.log_mid = (uint32_t)(log_n - pl.log_r[0] - pl.log_r[i]);
            launch_pass<P, true>(c, pl.log_r[i], blocks, (unsigned)nb, a);
        }
        ZKT_HIP(c, hipGetLastError());
    }
    return 0;
}

template <class P>
static int ntt_run_t(zkt_ctx* c, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out) {
    return ntt_run_batch_t<P>(c, log_n, inverse, coset, 1, &d_in, &in_len, &d_out);
}

int ntt_run_batch(zkt_ctx* c, int log_n, int inverse, int coset, int nb, const void* const* d_in, const size_t* in_len,
                  void* const* d_out) {
    if (c->curve == ZKT_CURVE_BN254) return ntt_run_batch_t<Bn254Fr>(c, log_n, inverse, coset, nb, d_in, in_len, d_out);
    return ntt_run_batch_t<Bls381Fr>(c, log_n, inverse, coset, nb, d_in, in_len, d_out);
}

int ntt_run(zkt_ctx* c, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out) {
    if (c->curve == ZKT_CURVE_BN254) return ntt_run_t<Bn254Fr>(c, log_n, inverse, coset, d_in, in_len, d_out);
    return ntt_run_t<Bls381Fr>(c, log_n, inverse, coset, d_in, in_len, d_out);
}

// p(X) mod (X^N - cN): out[i] = sum_k cN^k in[i + k N].  A polynomial longer than the class it is evaluated on
// (n + 8 blinded coefficients on a class of n or n / 2 points) is folded first; cN = shift^N.
template <class P>
__global__ void k_fold(const Fe<P>* in, uint64_t in_len, Fe<P>* out, uint64_t N, Fe<P> cN) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Fe<P> acc = (i < in_len) ? fe_load<P>(in + i) : fe_zero<P>();
    Fe<P> pw = cN;
    for (uint64_t at = i + N; at < in_len; at += N) {
        acc = fe_add<P>(acc, fe_mul<P>(pw, fe_load<P>(in + at)));
        pw = fe_mul<P>(pw, cN);
    }
    fe_store<P>(out + i, acc);
}

template <class P>
static int ntt_run_class_t(zkt_ctx* c, int log_n, int log_big, int cls, const void* d_in, size_t in_len, void* d_out,
                           void* d_fold) {
    const uint64_t N = (uint64_t)1 << log_n;
    if (log_big < log_n || log_big > P::TWO_ADICITY || cls < 0 || cls >= (1 << (log_big - log_n)) || cls > 255)
        return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "class transform: bad (log_n, log_big, class)");
    const int code = (log_big == log_n) ? 1 : ntt_class_code(log_big, cls);
    if (in_len > N) {
        if (!d_fold) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "class transform: fold scratch missing");
        const Fe<P> cN = fe_pow_u64<P>(coset_shift<P>(code), N);
        hipLaunchKernelGGL(k_fold<P>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (const Fe<P>*)d_in,
                           (uint64_t)in_len, (Fe<P>*)d_fold, N, cN);
        ZKT_HIP(c, hipGetLastError());
        d_in = d_fold;
        in_len = N;
    }
    return ntt_run_t<P>(c, log_n, 0, code, d_in, in_len, d_out);
}

// One GPU's share of the forward coset transform of size 2^log_big sharded by output index over G = 2^(log_big - log_n)
// GPUs: out[i] = p(g * w_big^(cls + G i)), i < 2^log_n.  in_len may exceed 2^log_n (d_fold: 2^log_n elements of scratch).
int ntt_run_class(zkt_ctx* c, int log_n, int log_big, int cls, const void* d_in, size_t in_len, void* d_out, void* d_fold) {
    if (c->curve == ZKT_CURVE_BN254) return ntt_run_class_t<Bn254Fr>(c, log_n, log_big, cls, d_in, in_len, d_out, d_fold);
    return ntt_run_class_t<Bls381Fr>(c, log_n, log_big, cls, d_in, in_len, d_out, d_fold);
}

}  // namespace zkt
