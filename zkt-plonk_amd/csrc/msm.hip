// KZG10 G1 multi-scalar multiplication on gfx950 (row a11 of SURVEY.md section 8).
//
// Replaces ark-ec 0.3 VariableBaseMSM::multi_scalar_mul as reached through
// SonicKZG10::commit / open (plonk-core/src/proof_system/prove.rs:133-135,178-180,249-251,306-308,
// 373-375,381-451) and HomomorphicCommitment::multi_scalar_mul (plonk-core/src/commitment.rs:32-45).
// Same group element, different schedule, chosen for HBM capacity and wave-wide execution:
//
//  * The SRS is fixed, so at load time every base gets its W = ceil((lambda+1)/c) window multiples
//    2^(c*w) * P_i precomputed in affine form ([w][i] table in HBM).  All windows then share ONE set
//    of 2^(c-1) buckets: there is no per-window bucket reduction and no Horner pass over windows.
//  * Signed c-bit digits halve the bucket count; a negative digit negates y on the fly.
//  * (bucket, table index) pairs are grouped by bucket with a two-level counting sort (below); accumulation walks
//    the grouped indices in fixed-size chunks (perfect lane balance whatever the digit distribution), emitting one
//    partial XYZZ sum per (chunk, bucket) piece at slot chunk + bucket; a second kernel folds a bucket's pieces.
//  * sum_b b * B_b goes through the row and column sums of the bucket matrix (one wavefront per sum), then one
//    single-wave workgroup per bit of b; the last doublings run on the host.
//  * The single resulting point is normalised (one inversion) on the host, which needs the affine
//    coordinates for the Fiat-Shamir transcript anyway.
//  * The prover's scalars are polynomial coefficients in arkworks' Montgomery form a = s R mod r.  The table holds
//    R^-1 P_i instead of P_i (scaled once at load), so sum a_i (R^-1 P_i) = sum s_i P_i and the digits are cut straight
//    from the words in memory: the into_repr() conversion (a Montgomery product per scalar in each of the two level-1
//    passes) is gone; canonical scalars (commitment.rs:36-42) are the ones that pay it now.
#include "ctx.hpp"
#include "ec.hpp"
#include "ecx.hpp"
#include "hostec.hpp"
#include "msm.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace zkt {

// ---------------------------------------------------------------------------------------------
// SRS: synthetic generation (test / bench trapdoor), window-multiple table
// ---------------------------------------------------------------------------------------------
// out[i] = tau^i * G  (insecure test SRS with a known trapdoor; PC::setup is out of scope)
template <class C>
__global__ void k_srs_generate(Affine<typename C::Fq>* out, size_t count, Fe<typename C::Fr> tau_mont,
                               Affine<typename C::Fq> g, size_t first) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // tau^i R^-1 as an integer: the table holds R^-1 P_i (see the header)
    Fe<R> s = fe_from_mont<R>(fe_from_mont<R>(fe_pow_u64<R>(tau_mont, (uint64_t)(first + i))));
    Xyzz<Q> acc = xyzz_identity<Q>();
    bool started = false;
#pragma unroll 1
    for (int li = R::N - 1; li >= 0; --li) {
        uint32_t e = s.v[li];
#pragma unroll 1
        for (int b = 31; b >= 0; --b) {
            if (started) acc = xyzz_double<Q>(acc);
            if ((e >> b) & 1u) {
                acc = xyzz_add_mixed<Q>(acc, g);
                started = true;
            }
        }
    }
    aff_store<Q>(out + i, xyzz_to_affine<Q>(acc));
}

// out[i] = [k] in[i] for one fixed scalar k (canonical words): k = R^-1 mod r when a caller's powers become the table's
// bases, k = R mod r on the way back out (zkt_srs_download).  FROM_FX: `in` is in the table's R' form.
template <class C, bool FROM_FX>
__global__ void k_srs_scale(const Affine<typename C::Fq>* in, Affine<typename C::Fq>* out, size_t count, Fe<typename C::Fr> k) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Affine<Q> base = aff_load<Q>(in + i);
    if (aff_is_inf<Q>(base)) {
        aff_store<Q>(out + i, base);
        return;
    }
    if (FROM_FX) {
        base.x = fx_to_ark<Q>(fx_unpack<Q>(base.x));
        base.y = fx_to_ark<Q>(fx_unpack<Q>(base.y));
    }
    Xyzz<Q> acc = xyzz_identity<Q>();
    bool started = false;
#pragma unroll 1
    for (int li = R::N - 1; li >= 0; --li) {
        uint32_t e = k.v[li];
#pragma unroll 1
        for (int b = 31; b >= 0; --b) {
            if (started) acc = xyzz_double<Q>(acc);
            if ((e >> b) & 1u) {
                acc = xyzz_add_mixed<Q>(acc, base);
                started = true;
            }
        }
    }
    aff_store<Q>(out + i, xyzz_to_affine<Q>(acc));
}

// table[w][i] = 2^(c*w) * table[0][i]
template <class C>
__global__ void k_srs_windows(Affine<typename C::Fq>* table, size_t count, MsmWindows win) {
    const int W = win.W;
    using Q = typename C::Fq;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Affine<Q> base = aff_load<Q>(table + i);
    if (aff_is_inf<Q>(base)) {
        for (int w = 1; w < W; ++w) aff_store<Q>(table + (size_t)w * count + i, base);
        return;
    }
    Xyzz<Q> acc = xyzz_from_affine<Q>(base);
#pragma unroll 1
    for (int w = 1; w < W; ++w) {
#pragma unroll 1
        for (int k = 0; k < (int)win.width[w - 1]; ++k) acc = xyzz_double<Q>(acc);
        Affine<Q> a = xyzz_to_affine<Q>(acc);
        aff_store<Q>(table + (size_t)w * count + i, a);
        acc = xyzz_from_affine<Q>(a);
    }
}

// ---------------------------------------------------------------------------------------------
// digits
// ---------------------------------------------------------------------------------------------
// signed digits of one scalar: emit(w, d, neg) with d in [0, 2^(width[w]-1)], d = 0 meaning "no contribution"
template <class R, class F>
ZKT_D void msm_for_each_digit(const Fe<R>& s, const MsmWindows& win, F&& emit) {
    const int W = win.W;
    uint64_t buf = 0;
    int bits = 0, w = 0;
    uint32_t carry = 0;
    auto one = [&](uint32_t raw, int c) {
        const uint32_t half = 1u << (c - 1);
        uint32_t d = raw + carry, neg = 0;
        if (d > half) {
            d = (1u << c) - d;
            neg = 1u;
            carry = 1u;
        } else {
            carry = 0u;
        }
        emit(w, d, neg);
        ++w;
    };
#pragma unroll
    for (int li = 0; li < R::N; ++li) {
        buf |= (uint64_t)s.v[li] << bits;
        bits += 32;
        while (w < W && bits >= (int)win.width[w]) {
            const int c = win.width[w];
            one((uint32_t)buf & ((1u << c) - 1u), c);
            buf >>= c;
            bits -= c;
        }
    }
    while (w < W) {
        const int c = win.width[w];
        one((uint32_t)buf & ((1u << c) - 1u), c);
        buf >>= c;
    }
}

// The same digits with the window layout known at compile time (W windows of LO bits, the first REM one bit wider):
// every shift, mask and word index folds to a constant and the scalar loads / branches of the generic loop disappear
// (the level-1 kernels spent as many scalar as vector instructions there).  DIG selects the layout: the two the
// prover's sizes produce (n = 2^19 .. 2^22 on either curve) have their own instance, everything else runs generic.
template <int DIG> struct DigitLayout { static constexpr int LO = 0, REM = 0, W = 0; };
template <> struct DigitLayout<1> { static constexpr int LO = 17, REM = 0, W = 15; };   // 255 bits = 15 x 17
template <> struct DigitLayout<2> { static constexpr int LO = 17, REM = 1, W = 15; };   // 256 bits = 18 + 14 x 17
template <> struct DigitLayout<3> { static constexpr int LO = 18, REM = 4, W = 14; };   // 256 bits = 4 x 19 + 10 x 18 (BLS12-381, n = 2^20)
template <> struct DigitLayout<4> { static constexpr int LO = 19, REM = 9, W = 13; };   // 256 bits = 9 x 20 + 4 x 19 (BLS12-381, n = 2^22)
template <> struct DigitLayout<5> { static constexpr int LO = 18, REM = 3, W = 14; };   // 255 bits = 3 x 19 + 11 x 18 (BN254, c = 19)
constexpr int MSM_DIGIT_LAYOUTS = 5;
static int msm_digit_layout(const MsmWindows& win, int total_bits) {
    static const int los[MSM_DIGIT_LAYOUTS + 1] = {0, DigitLayout<1>::LO, DigitLayout<2>::LO, DigitLayout<3>::LO, DigitLayout<4>::LO, DigitLayout<5>::LO};
    static const int rems[MSM_DIGIT_LAYOUTS + 1] = {0, DigitLayout<1>::REM, DigitLayout<2>::REM, DigitLayout<3>::REM, DigitLayout<4>::REM, DigitLayout<5>::REM};
    static const int ws[MSM_DIGIT_LAYOUTS + 1] = {0, DigitLayout<1>::W, DigitLayout<2>::W, DigitLayout<3>::W, DigitLayout<4>::W, DigitLayout<5>::W};
    for (int dig = 1; dig <= MSM_DIGIT_LAYOUTS; ++dig) {
        const int lo = los[dig], rem = rems[dig], W = ws[dig];
        if (win.W != W || W * lo + rem != total_bits) continue;
        bool ok = true;
        for (int w = 0; w < W; ++w) ok = ok && win.width[w] == lo + (w < rem ? 1 : 0);
        if (ok) return dig;
    }
    return 0;
}
template <class R, int DIG, class F>
ZKT_D void msm_for_each_digit_sel(const Fe<R>& s, const MsmWindows& win, F&& emit) {
    if constexpr (DIG == 0) {
        msm_for_each_digit<R>(s, win, emit);
    } else {
        constexpr int LO = DigitLayout<DIG>::LO, REM = DigitLayout<DIG>::REM, W = DigitLayout<DIG>::W;
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int start = w * LO + (w < REM ? w : REM), c = LO + (w < REM ? 1 : 0);
            const int word = start >> 5, sh = start & 31;
            uint64_t v = (word < R::N) ? s.v[word] : 0u;
            if (word + 1 < R::N) v |= (uint64_t)s.v[word + 1] << 32;
            const uint32_t raw = (uint32_t)(v >> sh) & ((1u << c) - 1u);   // bits at and above 32 N are zero
            const uint32_t half = 1u << (c - 1);
            uint32_t d = raw + carry, neg = 0;
            if (d > half) {
                d = (1u << c) - d;
                neg = 1u;
                carry = 1u;
            } else {
                carry = 0u;
            }
            emit(w, d, neg);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// grouping the (bucket, table index) pairs by bucket: a two-level counting sort made for this key
// shape (at most 18 key bits, roughly uniform digits) instead of a general radix sort.
//   level 1  splits on key >> 8 straight from the scalars: count, scan, scatter.  The digits are
//            recomputed rather than stored, and each workgroup orders its pairs in LDS first so that
//            the global writes are runs, not single pairs.
//   level 2  finishes every bin (256 buckets) in tiles of 8192 pairs: count, scan, scatter again,
//            the last one writing only the table indices plus offsets[bucket].
// Order inside a bucket is arbitrary, which is all the accumulation needs.  Zero digits are dropped.
// ---------------------------------------------------------------------------------------------
constexpr int MSM_BIN_LB_MAX = 8;         // level-2 key bits with 256-column level-2 tables (up to 2^17 buckets)
constexpr int MSM_BIN_LB_WIDE = 10;       // ... with 1024-column tables (more buckets: wide pairs only)
constexpr int MSM_L1_CAP = 15360;         // 8-byte pairs staged per level-1 workgroup (120 KB of LDS); twice as many 4-byte ones
// Pair formats after the level-1 split.  When the table index (< W * count), the sign and `lb` low key bits fit 32 bits
// the pair is ONE word: low key | index << lb | sign << 31 (n = 2^20: 24 index bits, lb = 7, 513 bins); the level-1
// scatter then writes, and both level-2 passes read, half the bytes.  Otherwise (larger keys) the pair is (key, value).
struct PairPacked {
    typedef uint32_t type;
    static __device__ __forceinline__ type make(uint32_t d, uint32_t idx, uint32_t neg, uint32_t lb) {
        return (d & ((1u << lb) - 1u)) | (idx << lb) | (neg << 31);
    }
    static __device__ __forceinline__ uint32_t low(type v, uint32_t lb) { return v & ((1u << lb) - 1u); }
    static __device__ __forceinline__ uint32_t val(type v, uint32_t lb) { return ((v & 0x7fffffffu) >> lb) | (v & 0x80000000u); }
};
struct PairWide {
    typedef uint2 type;
    static __device__ __forceinline__ type make(uint32_t d, uint32_t idx, uint32_t neg, uint32_t) { return make_uint2(d, idx | (neg << 31)); }
    static __device__ __forceinline__ uint32_t low(type v, uint32_t lb) { return v.x & ((1u << lb) - 1u); }
    static __device__ __forceinline__ uint32_t val(type v, uint32_t) { return v.y; }
};
constexpr int MSM_L2_TILE = 8192;         // pairs per level-2 workgroup
constexpr int MSM_MAX_NB1 = 1024;         // level-1 bins (one per thread in the scans below)

// exclusive scan of one value per thread over a 1024-thread workgroup; `wsum`: 16 LDS words
ZKT_D uint32_t block_excl_scan_1024(uint32_t mine, uint32_t* wsum, uint32_t* total) {
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    __syncthreads();   // wsum may still be read from a previous call
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
        const uint32_t v = wsum[w];
        if (w < (threadIdx.x >> 6)) before += v;
        all += v;
    }
    if (total) *total = all;
    return before + incl - mine;
}

// exclusive scan of one value per thread over the FIRST 256 threads of a workgroup (any block size that is a multiple of
// 64 and at least 256; every thread calls it); `wsum`: 4 LDS words.  Two barriers instead of the sixteen of a stepwise
// scan: these kernels are chains of latencies, not of arithmetic.
ZKT_D uint32_t block_excl_scan_256(uint32_t mine, uint32_t* wsum) {
    const bool in = threadIdx.x < 256;
    uint32_t incl = in ? mine : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    __syncthreads();   // wsum may still be read from a previous call
    if (in && (threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6) && w < 4; ++w) before += wsum[w];
    return before + incl - (in ? mine : 0u);
}

template <class C, int DIG>
__global__ __launch_bounds__(1024) void k_msm_bin_count(MsmBatch bt, int mont,
                                                        MsmWindows win, uint32_t per_block, uint32_t nb1, uint32_t lb,
                                                        uint32_t* counts) {
    using R = typename C::Fr;
    const uint32_t y = blockIdx.y;
    const Fe<R>* scalars = (const Fe<R>*)bt.scalars[y];
    const size_t n = bt.n[y];
    counts += y * bt.s_bin_offs;
    extern __shared__ uint32_t lds[];
    uint32_t* hist = lds;
    for (uint32_t b = threadIdx.x; b < nb1; b += 1024) hist[b] = 0;
    __syncthreads();
    // per_block <= 2048: at most two scalars per thread, both fetched before either is processed (these kernels wait
    // for memory most of their life, rocprof SQ_WAIT_ANY: every independent load in flight counts)
    Fe<R> sc[2];
    bool have[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t k = threadIdx.x + 1024u * t;
        const size_t i = (size_t)blockIdx.x * per_block + k;
        have[t] = k < per_block && i < n;
        if (have[t]) sc[t] = fe_load<R>(scalars + i);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (!have[t]) continue;
        Fe<R> s = sc[t];
        if (!mont) s = fe_to_mont<R>(s);   // the table holds R^-1 P_i: digits are those of s R
        msm_for_each_digit_sel<R, DIG>(s, win, [&](int, uint32_t d, uint32_t) {
            if (d) atomicAdd(&hist[d >> lb], 1u);
        });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += 1024) counts[(size_t)b * gridDim.x + blockIdx.x] = hist[b];
}

// exclusive scan of the level-1 counts in two small launches: 4096-element tiles scanned in place, then the
// tile totals (aux).  Readers add the two parts themselves: off(i) = counts[i] + aux[i >> 12].
constexpr int MSM_SCAN_TILE = 4096;
__global__ __launch_bounds__(1024) void k_msm_scan_tiles(uint32_t* counts, uint32_t total, uint32_t* aux, MsmBatch bt) {
    __shared__ uint32_t wsum[16];
    counts += blockIdx.y * bt.s_bin_offs;
    aux += blockIdx.y * bt.s_bin_aux;
    const uint32_t i0 = blockIdx.x * MSM_SCAN_TILE + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < total) ? counts[i0 + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t all;
    uint32_t run = block_excl_scan_1024(mine, wsum, &all);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < total) counts[i0 + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) aux[blockIdx.x] = all;
}
ZKT_D uint32_t msm_bin_off(const uint32_t* offs, const uint32_t* aux, size_t i, size_t total) {
    return (i < total) ? offs[i] + aux[i / MSM_SCAN_TILE] : aux[(total + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE];
}
// One workgroup: aux[0 .. nt) -> exclusive, aux[nt] = grand total; then the level-2 work list (bin b owns tiles
// [tile_start[b], tile_start[b + 1]) of MSM_L2_TILE pairs each, bin_start[b] = first pair of bin b); also resets the
// crowded-bucket counter of this MSM (read by k_msm_bucket_sum / k_msm_heavy later on the same stream).
__global__ __launch_bounds__(1024) void k_msm_scan_aux(const uint32_t* offs, uint32_t* aux, uint32_t nt, uint32_t nblk,
                                                       uint32_t nb1, uint32_t* bin_start, uint32_t* tile_start,
                                                       uint2* tile_desc, uint32_t acc_threads, uint32_t B, uint32_t lanes_fold,
                                                       MsmBatch bt) {
    __shared__ uint32_t wsum[16];
    const uint32_t y = blockIdx.y;
    offs += y * bt.s_bin_offs;
    aux += y * bt.s_bin_aux;
    bin_start += y * bt.s_bin;
    tile_start += y * bt.s_bin;
    tile_desc += y * bt.s_tile_desc;
    uint32_t* heavy_count = bt.heavy[y];
    uint32_t* params = bt.params[y];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nt; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t mine = (i < nt) ? aux[i] : 0u;
        uint32_t all;
        const uint32_t ex = block_excl_scan_1024(mine, wsum, &all);
        if (i < nt) aux[i] = carry + ex;
        carry += all;
    }
    if (threadIdx.x == 0) {
        aux[nt] = carry;
        *heavy_count = 0;
        // the accumulation's chunk (msm.hpp MSM_CHUNK_MIN): carry = pairs with a non-zero digit
        // lanes_fold (small keys): sixteen lanes fold a bucket's pieces, so the fold's chain is pieces / 16 + 4 additions and the
        // balance moves to much shorter chunks: isqrt(pairs per bucket / 11) instead of isqrt(2 pairs per bucket)
        uint32_t lo = 1;
        const uint32_t twice = lanes_fold ? (carry / B) / 11u : 2u * (carry / B);
        while (lo < (uint32_t)MSM_CHUNK_MIN && (lo + 1) * (lo + 1) <= twice) ++lo;
        const uint32_t even = (uint32_t)(((uint64_t)carry + acc_threads - 1) / acc_threads);
        params[0] = even > lo ? even : lo;
        params[1] = carry;
    }
    __syncthreads();   // aux is complete and visible to this workgroup
    const size_t total = (size_t)nb1 * nblk;
    uint32_t s = 0, tiles = 0;
    if (threadIdx.x <= nb1) s = msm_bin_off(offs, aux, (size_t)threadIdx.x * nblk, total);
    if (threadIdx.x < nb1) {
        const uint32_t e = msm_bin_off(offs, aux, (size_t)(threadIdx.x + 1) * nblk, total);
        tiles = (e - s + MSM_L2_TILE - 1) / MSM_L2_TILE;
    }
    uint32_t all;
    const uint32_t ex = block_excl_scan_1024(tiles, wsum, &all);
    if (threadIdx.x <= nb1) {
        bin_start[threadIdx.x] = s;
        tile_start[threadIdx.x] = ex;   // thread nb1 contributes 0 tiles, so this is the grand total there
    }
    // [first pair, one past the last pair) of every level-2 tile: the level-2 kernels read ONE descriptor instead of
    // searching tile_start (ten dependent loads at the head of a workgroup that lives for ten microseconds)
    if (threadIdx.x < nb1) {
        const uint32_t be = msm_bin_off(offs, aux, (size_t)(threadIdx.x + 1) * nblk, total);
        for (uint32_t k = 0; k < tiles; ++k) {
            const uint32_t ts = s + k * MSM_L2_TILE;
            tile_desc[ex + k] = make_uint2(ts, (ts + MSM_L2_TILE < be) ? ts + MSM_L2_TILE : be);
        }
    }
}

// LDS: cursor[nb1] | delta[nb1] | first[nb1] | stage PF::type[...]
// Every workgroup orders its pairs by bin in LDS (positions from LDS atomics on the bin cursors), then one wavefront
// per bin copies the bin's run to its place in `pairs`: 64 consecutive entries per instruction.
template <class C, class PF, int DIG>
__global__ __launch_bounds__(1024) void k_msm_bin_scatter(MsmBatch bt, int mont,
                                                          MsmWindows win, uint32_t per_block,
                                                          uint32_t nb1, uint32_t lb, const uint32_t* offs,
                                                          const uint32_t* aux, typename PF::type* pairs) {
    using R = typename C::Fr;
    const uint32_t y = blockIdx.y;
    const Fe<R>* scalars = (const Fe<R>*)bt.scalars[y];
    const size_t n = bt.n[y], base_off = bt.base_off[y], count = bt.tcount[y];
    offs += y * bt.s_bin_offs;
    aux += y * bt.s_bin_aux;
    pairs = reinterpret_cast<typename PF::type*>(reinterpret_cast<char*>(pairs) + y * bt.s_pairs_bytes);
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wsum[16];
    uint32_t* cursor = lds;
    uint32_t* delta = lds + nb1;
    uint32_t* first = lds + 2 * nb1;
    typename PF::type* stage = (typename PF::type*)(lds + ((3 * nb1 + 3) & ~3u));
    const size_t total = (size_t)nb1 * gridDim.x;
    // this workgroup's level-1 histogram is the difference of neighbouring scanned counts
    uint32_t g0 = 0, mine = 0;
    if (threadIdx.x < nb1) {
        const size_t at = (size_t)threadIdx.x * gridDim.x + blockIdx.x;
        g0 = msm_bin_off(offs, aux, at, total);
        mine = msm_bin_off(offs, aux, at + 1, total) - g0;
    }
    uint32_t staged;
    const uint32_t ex = block_excl_scan_1024(mine, wsum, &staged);
    if (threadIdx.x < nb1) {
        cursor[threadIdx.x] = ex;
        first[threadIdx.x] = ex;
        delta[threadIdx.x] = g0 - ex;
    }
    __syncthreads();
    Fe<R> sc[2];
    bool have[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // per_block <= 2048 (see k_msm_bin_count)
        const uint32_t k = threadIdx.x + 1024u * t;
        const size_t i = (size_t)blockIdx.x * per_block + k;
        have[t] = k < per_block && i < n;
        if (have[t]) sc[t] = fe_load<R>(scalars + i);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (!have[t]) continue;
        const size_t i = (size_t)blockIdx.x * per_block + threadIdx.x + 1024u * t;
        Fe<R> s = sc[t];
        if (!mont) s = fe_to_mont<R>(s);
        msm_for_each_digit_sel<R, DIG>(s, win, [&](int w, uint32_t d, uint32_t neg) {
            if (d) {
                const uint32_t at = atomicAdd(&cursor[d >> lb], 1u);
                stage[at] = PF::make(d, (uint32_t)((size_t)w * count + base_off + i), neg, lb);
            }
        });
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = threadIdx.x >> 6; b < nb1; b += 16) {
        const uint32_t s0 = first[b], e0 = cursor[b], d = delta[b];
        for (uint32_t j = s0 + lane; j < e0; j += 64) pairs[d + j] = stage[j];
    }
}

struct L2Item {
    uint32_t s, e;
    bool valid;
};
ZKT_D L2Item msm_l2_item(uint32_t item, uint32_t nb1, const uint32_t* tile_start, const uint2* tile_desc) {
    L2Item r;
    r.valid = item < tile_start[nb1];
    r.s = 0; r.e = 0;
    if (!r.valid) return r;
    const uint2 d = tile_desc[item];
    r.s = d.x;
    r.e = d.y;
    return r;
}

// LC = log2 of the level-2 table's columns: 8 for up to 2^17 buckets (256 columns, lb <= 8), 10 beyond (1024 columns,
// lb = 10: the digit widths c = 19, 20 that pay at n >= 2^22, where W n additions outweigh 2^(c-1) buckets to reduce).
template <class PF, int LC>
__global__ __launch_bounds__(256) void k_msm_l2_count(const typename PF::type* pairs, uint32_t nb1, uint32_t lb,
                                                      const uint32_t* tile_start, const uint2* tile_desc, uint32_t* cnt2,
                                                      MsmBatch bt) {
    constexpr uint32_t COLS = 1u << LC;
    {
        const uint32_t y = blockIdx.y;
        pairs = reinterpret_cast<const typename PF::type*>(reinterpret_cast<const char*>(pairs) + y * bt.s_pairs_bytes);
        tile_start += y * bt.s_bin;
        tile_desc += y * bt.s_tile_desc;
        cnt2 += y * bt.s_cnt;
    }
    __shared__ uint32_t hist[COLS];
    const L2Item it = msm_l2_item(blockIdx.x, nb1, tile_start, tile_desc);
    if (!it.valid) return;
#pragma unroll
    for (uint32_t c = threadIdx.x; c < COLS; c += 256) hist[c] = 0;
    __syncthreads();
    // eight independent loads per thread in flight before the first LDS atomic
    const uint32_t cnt = it.e - it.s;
    for (uint32_t base = 0; base < cnt; base += 256u * 8u) {
        typename PF::type v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t j = base + 256u * k + threadIdx.x;
            if (j < cnt) v[k] = pairs[it.s + j];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t j = base + 256u * k + threadIdx.x;
            if (j < cnt) atomicAdd(&hist[PF::low(v[k], lb)], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = threadIdx.x; c < COLS; c += 256) cnt2[(size_t)blockIdx.x * COLS + c] = hist[c];
}

// one workgroup per bin: positions of every (tile, bucket) run and offsets[bucket].  Thread t owns the K = COLS / 256
// consecutive columns t K .. t K + K - 1.
template <int LC>
__global__ __launch_bounds__(256) void k_msm_l2_scan(const uint32_t* cnt2, uint32_t* pos2, const uint32_t* bin_start,
                                                     const uint32_t* tile_start, uint32_t B, uint32_t lb,
                                                     uint32_t* chunk_bucket, MsmBatch bt) {
    constexpr uint32_t COLS = 1u << LC, K = COLS / 256u;
    const uint32_t y = blockIdx.y;
    cnt2 += y * bt.s_cnt;
    pos2 += y * bt.s_cnt;
    bin_start += y * bt.s_bin;
    tile_start += y * bt.s_bin;
    chunk_bucket += y * bt.s_chunk;
    uint32_t* offsets = bt.offsets[y];
    const uint32_t chunk = bt.params[y][0];
    const uint32_t b = blockIdx.x;
    const uint32_t t0 = tile_start[b], t1 = tile_start[b + 1];
    uint32_t run[K];
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) run[k] = 0;
    for (uint32_t t = t0; t < t1; ++t) {
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) {
            const size_t at = (size_t)t * COLS + threadIdx.x * K + k;
            pos2[at] = run[k];
            run[k] += cnt2[at];
        }
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) mine += run[k];
    __shared__ uint32_t wsum4[4];
    uint32_t first[K];
    first[0] = bin_start[b] + block_excl_scan_256(mine, wsum4);
#pragma unroll
    for (uint32_t k = 1; k < K; ++k) first[k] = first[k - 1] + run[k - 1];
    // accumulation chunk t starts at pair t * chunk: tell it which bucket that pair belongs to.  A bucket normally
    // covers a handful of chunks; a crowded one (skewed digits) is written by the whole workgroup.
    __shared__ uint32_t big[256][3];
    __shared__ uint32_t nbig;
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < K; ++k) {
        // bucket = (bin << lb) | low; columns at or above 2^lb never receive a pair and own no bucket
        const uint32_t col = threadIdx.x * K + k;
        const bool owns = col < (1u << lb);
        const uint32_t key = (b << lb) + col;
        if (owns) offsets[key] = first[k];   // sized (nb1 << lb) + 2; keys above B are empty and repeat the end
        if (owns && key >= 1 && key <= B && run[k]) {
            const uint32_t tb = (first[k] + chunk - 1) / chunk;
            const uint32_t te = (uint32_t)(((uint64_t)first[k] + run[k] + chunk - 1) / chunk);   // one past the last chunk start inside
            uint32_t at = 256;
            if (te - tb > 8) at = atomicAdd(&nbig, 1u);   // (small keys cut a bucket into dozens of four-pair chunks)
            if (at < 256) {
                big[at][0] = key; big[at][1] = tb; big[at][2] = te;
            } else {
                for (uint32_t t = tb; t < te; ++t) chunk_bucket[t] = key;
            }
        }
    }
    __syncthreads();
    const uint32_t nb = nbig < 256u ? nbig : 256u;
    for (uint32_t e = 0; e < nb; ++e)
        for (uint32_t t = big[e][1] + threadIdx.x; t < big[e][2]; t += 256) chunk_bucket[t] = big[e][0];
    for (uint32_t t = t0; t < t1; ++t) {
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) pos2[(size_t)t * COLS + threadIdx.x * K + k] += first[k];
    }
}

// 1024 threads per 8192-pair tile: the 42 KB of LDS (56 KB with 1024 columns) allow two such workgroups (32 waves) per CU,
// where 256-thread workgroups left 12 waves to hide the latency this kernel consists of.  The pair loads are issued before
// the scan of the counters, so that they travel while it runs.
constexpr int MSM_L2S_THREADS = 1024;
constexpr int MSM_L2S_PER = MSM_L2_TILE / MSM_L2S_THREADS;   // pairs per thread
template <class PF, int LC>
__global__ __launch_bounds__(MSM_L2S_THREADS) void k_msm_l2_scatter(const typename PF::type* pairs, uint32_t nb1, uint32_t lb,
                                                                    const uint32_t* tile_start, const uint2* tile_desc,
                                                                    const uint32_t* cnt2, const uint32_t* pos2,
                                                                    uint32_t* vals, MsmBatch bt) {
    constexpr uint32_t COLS = 1u << LC;
    {
        const uint32_t y = blockIdx.y;
        pairs = reinterpret_cast<const typename PF::type*>(reinterpret_cast<const char*>(pairs) + y * bt.s_pairs_bytes);
        tile_start += y * bt.s_bin;
        tile_desc += y * bt.s_tile_desc;
        cnt2 += y * bt.s_cnt;
        pos2 += y * bt.s_cnt;
        vals += y * bt.s_vals;
    }
    typedef typename std::conditional<(LC <= 8), uint8_t, uint16_t>::type key_t;
    __shared__ uint32_t cursor[COLS], delta[COLS], wsum[16];
    __shared__ uint32_t sval[MSM_L2_TILE];
    __shared__ key_t skey[MSM_L2_TILE];
    const L2Item it = msm_l2_item(blockIdx.x, nb1, tile_start, tile_desc);
    if (!it.valid) return;
    const uint32_t cnt = it.e - it.s;
    typename PF::type v[MSM_L2S_PER];
#pragma unroll
    for (int k = 0; k < MSM_L2S_PER; ++k) {
        const uint32_t j = (uint32_t)MSM_L2S_THREADS * k + threadIdx.x;
        if (j < cnt) v[k] = pairs[it.s + j];
    }
    uint32_t mine = 0, p2 = 0;
    if (threadIdx.x < COLS) {
        mine = cnt2[(size_t)blockIdx.x * COLS + threadIdx.x];
        p2 = pos2[(size_t)blockIdx.x * COLS + threadIdx.x];
    }
    uint32_t ex;
    if constexpr (LC <= 8) ex = block_excl_scan_256(mine, wsum);
    else ex = block_excl_scan_1024(mine, wsum, nullptr);
    if (threadIdx.x < COLS) {
        cursor[threadIdx.x] = ex;
        delta[threadIdx.x] = p2 - ex;
    }
    __syncthreads();
    uint32_t a[MSM_L2S_PER];
#pragma unroll
    for (int k = 0; k < MSM_L2S_PER; ++k) {
        const uint32_t j = (uint32_t)MSM_L2S_THREADS * k + threadIdx.x;
        if (j < cnt) a[k] = atomicAdd(&cursor[PF::low(v[k], lb)], 1u);
    }
#pragma unroll
    for (int k = 0; k < MSM_L2S_PER; ++k) {
        const uint32_t j = (uint32_t)MSM_L2S_THREADS * k + threadIdx.x;
        if (j < cnt) {
            sval[a[k]] = PF::val(v[k], lb);
            skey[a[k]] = (key_t)PF::low(v[k], lb);
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < cnt; j += MSM_L2S_THREADS) vals[delta[skey[j]] + j] = sval[j];
}

// ---------------------------------------------------------------------------------------------
// accumulation: fixed-size chunks over the grouped table indices, one piece per (chunk, bucket).
// Coordinates live in registers as lazily reduced 29-bit limbs (ecx.hpp); the table and the buckets are canonical
// packed words in R' Montgomery form, the pieces are the raw limbs (XyzzRaw).
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256) void k_msm_accumulate(const uint32_t* vals, uint32_t B, const uint32_t* chunk_bucket,
                                                        MsmBatch bt) {
    using Q = typename C::Fq;
    const uint32_t y = blockIdx.y;
    const Affine<Q>* table = (const Affine<Q>*)bt.table[y];
    vals += y * bt.s_vals;
    chunk_bucket += y * bt.s_chunk;
    const uint32_t* offsets = bt.offsets[y];
    XyzzRaw<Q>* pieces = (XyzzRaw<Q>*)bt.pieces[y];
    const uint32_t chunk = bt.params[y][0];
    const uint32_t base = offsets[1], m = offsets[B + 1];   // first / one past the last pair with a non-zero bucket
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = (uint64_t)base + (uint64_t)t * chunk;
    if (p0 >= m) return;
    const uint32_t p1 = (uint32_t)((p0 + chunk < m) ? p0 + chunk : m);
    uint32_t cur = chunk_bucket[t];   // bucket of the first pair (k_msm_l2_scan): offsets[cur] <= p0 < offsets[cur + 1]
    // end of the current bucket and of the next one: the second is fetched a whole bucket ahead, so that moving on
    // to the next bucket does not wait for memory (only runs of empty buckets do)
    uint32_t end = offsets[cur + 1];
    uint32_t end2 = offsets[(cur + 2 <= B + 1) ? cur + 2 : B + 1];
    XyzzX<Q> acc = xx_identity<Q>();
    // two-deep software pipeline: the index of pair p+2 and the point of pair p+1 are fetched behind the addition of
    // pair p, so that no load in the loop waits for another load
    uint32_t v = vals[p0];
    uint32_t v1 = (p0 + 1 < p1) ? vals[p0 + 1] : 0u;
    Fe<Q> nx = fe_load<Q>(&table[v & 0x7fffffffu].x), ny = fe_load<Q>(&table[v & 0x7fffffffu].y);
    for (uint32_t p = (uint32_t)p0; p < p1; ++p) {
        const uint32_t vcur = v;
        const Fe<Q> cx = nx, cy = ny;
        v = v1;
        if (p + 1 < p1) {
            nx = fe_load<Q>(&table[v & 0x7fffffffu].x);
            ny = fe_load<Q>(&table[v & 0x7fffffffu].y);
        }
        if (p + 2 < p1) v1 = vals[p + 2];
        if (p == end) {    // next non-empty bucket
            xx_store_raw<Q>(pieces + (size_t)t + cur, acc);
            acc = xx_identity<Q>();
            do {
                ++cur;
                end = end2;
                end2 = offsets[(cur + 2 <= B + 1) ? cur + 2 : B + 1];
            } while (end <= p);
        }
        if (!(fe_is_zero<Q>(cx) && fe_is_zero<Q>(cy))) {
            AffineX<Q> q;
            q.x = fx_unpack<Q>(cx);
            q.y = fx_unpack<Q>(cy);
            if (vcur >> 31) q.y = fx_sub<Q, 1>(fx_zero<Q>(), q.y);  // -y = p - y
            acc = xx_add_mixed<Q, true>(acc, q);
        }
    }
    xx_store_raw<Q>(pieces + (size_t)t + cur, acc);
}

// MSM_TAIL_INL (template parameter of the tail kernels): the bucket reduction's additions with inlined products (ecx.hpp xx_add)

template <class C, bool MSM_TAIL_INL>
__global__ __launch_bounds__(256) void k_msm_bucket_sum(uint32_t B, MsmTailBatch tb) {
    using Q = typename C::Fq;
    const uint32_t* offsets = tb.offsets[blockIdx.y];
    const XyzzRaw<Q>* pieces = (const XyzzRaw<Q>*)tb.pieces[blockIdx.y];
    Xyzz<Q>* buckets = (Xyzz<Q>*)tb.buckets[blockIdx.y];
    uint32_t* heavy = tb.heavy[blockIdx.y];
    const uint32_t chunk = tb.params[blockIdx.y][0];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;  // 0..B ; bucket 0 is the identity
    if (b > B) return;
    XyzzX<Q> acc = xx_identity<Q>();
    if (b >= 1) {
        const uint32_t base = offsets[1];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (e > s) {
            const uint32_t t0 = (s - base) / chunk, t1 = (e - 1 - base) / chunk;
            if (t1 - t0 >= MSM_HEAVY) {  // crowded bucket: leave it to k_msm_heavy
                heavy[1 + atomicAdd(heavy, 1u)] = b;
                return;
            }
            acc = xx_load_raw<Q>(pieces + (size_t)t0 + b);
            for (uint32_t t = t0 + 1; t <= t1; ++t) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load_raw<Q>(pieces + (size_t)t + b));
        }
    }
    xx_store<Q>(buckets + b, acc);
}

template <class Q>
ZKT_D XyzzX<Q> xx_shfl_down(const XyzzX<Q>& p, int delta);

// The same fold by sixteen lanes per bucket (small keys, where a proof is a chain of latencies and the chip is mostly idle):
// lane l of the group adds pieces t0 + l, t0 + l + 16, ..., then four shuffle steps; any number of pieces, no crowded-bucket
// list.  With it the accumulation's chunks can be four pairs long instead of sixteen (k_msm_scan_aux).
template <class C, bool MSM_TAIL_INL>
__global__ __launch_bounds__(256) void k_msm_bucket_sum_lanes(uint32_t B, MsmTailBatch tb) {
    using Q = typename C::Fq;
    const uint32_t* offsets = tb.offsets[blockIdx.y];
    const XyzzRaw<Q>* pieces = (const XyzzRaw<Q>*)tb.pieces[blockIdx.y];
    Xyzz<Q>* buckets = (Xyzz<Q>*)tb.buckets[blockIdx.y];
    const uint32_t chunk = tb.params[blockIdx.y][0];
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = g >> 4, lane = g & 15u;   // 0..B ; bucket 0 is the identity
    XyzzX<Q> acc = xx_identity<Q>();
    if (b >= 1 && b <= B) {
        const uint32_t base = offsets[1];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (e > s) {
            const uint32_t t0 = (s - base) / chunk, t1 = (e - 1 - base) / chunk;
            for (uint32_t t = t0 + lane; t <= t1; t += 16) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load_raw<Q>(pieces + (size_t)t + b));
        }
    }
#pragma unroll 1
    for (int d = 8; d >= 1; d >>= 1) {   // whole wavefronts take every step (b may exceed B in the last one: identities)
        XyzzX<Q> o = xx_shfl_down<Q>(acc, d);
        if ((int)lane + d >= 16) o = xx_identity<Q>();
        acc = xx_add<Q, MSM_TAIL_INL>(acc, o);
    }
    if (lane == 0 && b <= B) xx_store<Q>(buckets + b, acc);
}

template <class Q>
ZKT_D XyzzX<Q> xx_shfl_down(const XyzzX<Q>& p, int delta) {
    XyzzX<Q> r;
#pragma unroll
    for (int i = 0; i < FxP<Q>::L; ++i) {
        r.x.l[i] = __shfl_down(p.x.l[i], delta);
        r.y.l[i] = __shfl_down(p.y.l[i], delta);
        r.zz.l[i] = __shfl_down(p.zz.l[i], delta);
        r.zzz.l[i] = __shfl_down(p.zzz.l[i], delta);
    }
    r.inf = __shfl_down((int)p.inf, delta) != 0;
    return r;
}

// block-wide sum of one point per thread (256 threads); result valid in thread 0.  `wsum`: 4 LDS slots.
template <class Q, bool MSM_TAIL_INL>
ZKT_D XyzzX<Q> block_sum_256(XyzzX<Q> acc, Xyzz<Q>* wsum) {
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        XyzzX<Q> o = xx_shfl_down<Q>(acc, d);
        if ((threadIdx.x & 63) + d >= 64) o = xx_identity<Q>();
        acc = xx_add<Q, MSM_TAIL_INL>(acc, o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) xx_store<Q>(wsum + wv, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = xx_load<Q>(wsum);
        for (int i = 1; i < 4; ++i) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load<Q>(wsum + i));
    }
    __syncthreads();
    return acc;
}

// crowded buckets (skewed digit distributions): one block folds all pieces of one bucket
template <class C, bool MSM_TAIL_INL>
__global__ __launch_bounds__(256) void k_msm_heavy(MsmTailBatch tb) {
    using Q = typename C::Fq;
    const uint32_t* offsets = tb.offsets[blockIdx.y];
    const XyzzRaw<Q>* pieces = (const XyzzRaw<Q>*)tb.pieces[blockIdx.y];
    Xyzz<Q>* buckets = (Xyzz<Q>*)tb.buckets[blockIdx.y];
    const uint32_t* heavy = tb.heavy[blockIdx.y];
    const uint32_t chunk = tb.params[blockIdx.y][0];
    __shared__ Xyzz<Q> wsum[4];
    const uint32_t nheavy = heavy[0];
    const uint32_t base = offsets[1];
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy[1 + h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t t0 = (s - base) / chunk, t1 = (e - 1 - base) / chunk;
        XyzzX<Q> acc = xx_identity<Q>();
        for (uint32_t t = t0 + threadIdx.x; t <= t1; t += 256) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load_raw<Q>(pieces + (size_t)t + b));
        acc = block_sum_256<Q, MSM_TAIL_INL>(acc, wsum);
        if (threadIdx.x == 0) xx_store<Q>(buckets + b, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// bucket reduction  S = sum_{b=0..B} b * bucket[b] = 2^(c-1) * bucket[B] + sum_k 2^k * (sum of bucket[b] over b < B with bit k)
// through row and column sums: with b = i NJ + j (NJ = 2^q2 columns, NI = 2^q1 rows, q1 + q2 = c - 1) the sum of the
// buckets whose bit k is set equals the sum of the COLUMN sums C_j over the j with bit k (k < q2), or of the ROW sums
// R_i over the i with bit k - q2.  So the 2^(c-1) buckets are read twice to form NI + NJ sums (one wavefront each), and
// the bit-weighted sums run over 2^q1 + 2^q2 = 512 points instead of 65536.  r02 went through 4-bucket running sums
// and then read every segment once per bit (14 x 8192 additions on 120 workgroups that held registers the accumulation
// was waiting for): seven times the additions, four times the waves, ten dependent additions more.
// ---------------------------------------------------------------------------------------------
template <class Q, bool MSM_TAIL_INL>
ZKT_D XyzzX<Q> wave_sum(XyzzX<Q> acc) {   // valid in lane 0
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        XyzzX<Q> o = xx_shfl_down<Q>(acc, d);
        if ((threadIdx.x & 63) + d >= 64) o = xx_identity<Q>();
        acc = xx_add<Q, MSM_TAIL_INL>(acc, o);
    }
    return acc;
}

// wavefront w < NI: R_w = sum_j bucket[w NJ + j];  NI <= w < NI + NJ: C_(w - NI) = sum_i bucket[i NJ + (w - NI)]
template <class C, bool MSM_TAIL_INL>
__global__ __launch_bounds__(256) void k_msm_rowcol(uint32_t q1, uint32_t q2, MsmTailBatch tb) {
    using Q = typename C::Fq;
    const Xyzz<Q>* buckets = (const Xyzz<Q>*)tb.buckets[blockIdx.y];
    Xyzz<Q>* rc = (Xyzz<Q>*)tb.rowcol[blockIdx.y];
    const uint32_t NI = 1u << q1, NJ = 1u << q2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= NI + NJ) return;
    XyzzX<Q> acc = xx_identity<Q>();
    if (w < NI) {
        for (uint32_t j = lane; j < NJ; j += 64) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load<Q>(buckets + (size_t)w * NJ + j));
    } else {
        const uint32_t j = w - NI;
        for (uint32_t i = lane; i < NI; i += 64) acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load<Q>(buckets + (size_t)i * NJ + j));
    }
    acc = wave_sum<Q, MSM_TAIL_INL>(acc);
    if (lane == 0) xx_store<Q>(rc + w, acc);
}

// One single-wavefront workgroup per row of the host's table (sixteen of them on one CU took turns at the registers):
// y = k < q2: sum of C_j over j with bit k; y = q2 + k, k < q1: sum of R_i over i with bit k; y = q1 + q2: the top bucket
// alone.  Written in arkworks' R form straight into pinned host memory: the host applies the weights 2^y
// (hostec.hpp weighted_row_sum) -- the remaining ~35 dependent curve operations cost a wavefront 0.6 ms and the host 15 us.
template <class C, bool MSM_TAIL_INL>
__global__ __launch_bounds__(64) void k_msm_weighted_rows(uint32_t q1, uint32_t q2, uint32_t B, MsmTailBatch tb) {
    using Q = typename C::Fq;
    const Xyzz<Q>* rc = (const Xyzz<Q>*)tb.rowcol[blockIdx.y];
    const Xyzz<Q>* top_bucket = (const Xyzz<Q>*)tb.buckets[blockIdx.y] + B;
    Xyzz<Q>* partials = (Xyzz<Q>*)tb.partials[blockIdx.y];
    const uint32_t NI = 1u << q1, NJ = 1u << q2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t y = blockIdx.x;
    XyzzX<Q> acc = xx_identity<Q>();
    if (y == q1 + q2) {
        if (lane == 0) acc = xx_load<Q>(top_bucket);
    } else {
        const bool col = y < q2;
        const uint32_t k = col ? y : y - q2, low = (1u << k) - 1u;
        const uint32_t cnt = (col ? NJ : NI) >> 1;
        const Xyzz<Q>* src = col ? rc + NI : rc;
        for (uint32_t q = lane; q < cnt; q += 64) {
            const uint32_t idx = ((q & ~low) << 1) | (1u << k) | (q & low);   // q with a one inserted at bit k
            acc = xx_add<Q, MSM_TAIL_INL>(acc, xx_load<Q>(src + idx));
        }
    }
    acc = wave_sum<Q, MSM_TAIL_INL>(acc);
    if (lane == 0) xx_store_ark<Q>(partials + y, acc);
}

// table: arkworks R form -> R' form (canonical packed), in place; (0,0) stays (0,0)
template <class C>
__global__ void k_srs_to_fx(Affine<typename C::Fq>* table, size_t total) {
    using Q = typename C::Fq;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    Affine<Q> a = aff_load<Q>(table + i);
    if (aff_is_inf<Q>(a)) return;
    a.x = fx_pack<Q>(fx_cond_sub_p<Q>(fx_from_ark<Q>(a.x)));
    a.y = fx_pack<Q>(fx_cond_sub_p<Q>(fx_from_ark<Q>(a.y)));
    aff_store<Q>(table + i, a);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// rows of partial sums an MSM leaves for the host besides the top bucket's: one per bit of the bucket index below B
static int msm_rows(int c) { return c - 1; }

// level-1 kernels by digit layout (0 = generic loop)
template <class C>
static auto msm_pick_bin_count(int dig) -> decltype(&k_msm_bin_count<C, 0>) {
    switch (dig) {
        case 1: return k_msm_bin_count<C, 1>;
        case 2: return k_msm_bin_count<C, 2>;
        case 3: return k_msm_bin_count<C, 3>;
        case 4: return k_msm_bin_count<C, 4>;
        case 5: return k_msm_bin_count<C, 5>;
        default: return k_msm_bin_count<C, 0>;
    }
}
template <class C, class PF>
static auto msm_pick_bin_scatter(int dig) -> decltype(&k_msm_bin_scatter<C, PF, 0>) {
    switch (dig) {
        case 1: return k_msm_bin_scatter<C, PF, 1>;
        case 2: return k_msm_bin_scatter<C, PF, 2>;
        case 3: return k_msm_bin_scatter<C, PF, 3>;
        case 4: return k_msm_bin_scatter<C, PF, 4>;
        case 5: return k_msm_bin_scatter<C, PF, 5>;
        default: return k_msm_bin_scatter<C, PF, 0>;
    }
}

static int floor_log2(size_t x) {
    int l = 0;
    while ((x >> (l + 1)) != 0) ++l;
    return l;
}

template <class C>
static int msm_setup(zkt_ctx* c, size_t count, const MsmState* share = nullptr) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    auto st = std::make_shared<MsmState>();
    st->count = count;
    // Digit width.  W n additions against 2^(c-1) buckets to reduce and a sort whose pairs stop fitting 32 bits beyond
    // 2^17 buckets: measured on MI355X (profiles/ab_digit_width_r04.txt), BN254 n = 2^20 is best at c = 17 (wider digits
    // give the accumulation 10 % and take it back in the grouping), BLS12-381 -- whose additions cost 2.4 x as much, the
    // sort the same -- gains 4.6 % per proof from c = 19 at n = 2^20 and 7 % from c = 20 at n = 2^22.
    int lg = floor_log2(count ? count : 1);
    int cb = lg - 2;
    if (cb < 8) cb = 8;
    if (lg <= 20) {
        if (cb > 18) cb = 18;
        // r05 (profiles/ab_digit_width_r05.txt): BN254 wants its fifteen windows of 17 bits from n = 2^17 on (2^17 +6.4 %, 2^18 +8.2 %
        // per proof against lg - 2), BLS12-381 n = 2^19 fifteen of 18 (+1.3 %); below that the bucket reductions dominate
        if (Q::N == 12) {
            if (lg == 20) cb = 19;
            else if (lg == 19) cb = 18;
            else if (lg == 17) cb = 16;   // sixteen windows instead of eighteen: +18 %
        } else if (lg >= 17) {
            cb = 17;
        }
    } else if (cb > 20) {
        cb = 20;   // 2^19 buckets: the sort's limit (1024 level-1 bins x 1024 level-2 columns, wide pairs)
    }
    if (const char* e = exp_env("ZKT_MSM_CBITS")) {   // experiment: another digit width (fewer windows, more buckets)
        const int f = atoi(e);
        if (f >= 8 && f <= MSM_MAX_Y) cb = f;
    }
    {   // W windows of width cmax or cmax-1 covering exactly lambda+1 bits
        const int total = R::BITS + 1;
        const int W = (total + cb - 1) / cb;
        const int lo = total / W, rem = total % W;
        st->win.W = W;
        int pos = 0;
        for (int w = 0; w < W; ++w) {
            st->win.width[w] = (uint8_t)(lo + (w < rem ? 1 : 0));
            st->win.start[w] = (uint16_t)pos;
            pos += st->win.width[w];
        }
        st->W = W;
        st->c = lo + (rem ? 1 : 0);
        cb = st->c;
        st->dig = msm_digit_layout(st->win, total);
    }
    st->B = 1u << (cb - 1);
    if ((uint64_t)st->W * count >= ((uint64_t)1 << 31))
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "SRS too large for 31-bit table indices");
    int rc;
    if (share) {   // zkt_ctx_fork: the parent's tables (same count, hence the same layout), everything below this context's own
        st->table = share->table;
        st->table_borrowed = true;
        st->table2 = share->table2;
        st->table2_borrowed = share->table2 != nullptr;
        st->count2 = share->count2;
        st->lag_log_n = share->lag_log_n;
        st->lag_failed = share->lag_failed;
        st->slice_off = share->slice_off;
        st->total = share->total;
    } else if ((rc = dev_alloc(c, &st->table, (size_t)st->W * count * sizeof(Affine<Q>)))) {
        return rc;
    }
    size_t m = (size_t)st->W * count;
    // the main-stream work buffers exist MSM_BATCH times (a round's commitments are grouped and accumulated as one batch:
    // msm_enqueue_batch); strides between the copies in st->strides
    constexpr size_t NB = MSM_BATCH;
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };   // keep every copy 256-byte aligned
    st->strides.s_vals = up(m);
    st->strides.s_pairs_bytes = up(m) * 8;
    if ((rc = dev_alloc(c, (void**)&st->vals2, NB * st->strides.s_vals * 4))) return rc;
    if ((rc = dev_alloc(c, &st->pairs, NB * st->strides.s_pairs_bytes))) return rc;
    {   // pair format: one 32-bit word when low key bits + table index + sign fit (see PairPacked)
        int idx_bits = 1;
        while ((m - 1) >> idx_bits) ++idx_bits;
        int lb = 31 - idx_bits;
        if (lb > MSM_BIN_LB_MAX) lb = MSM_BIN_LB_MAX;
        st->packed = lb >= 4 && ((st->B >> lb) + 1) < (uint32_t)MSM_MAX_NB1;
        st->lb = st->packed ? (uint32_t)lb : (uint32_t)MSM_BIN_LB_MAX;
        if (!st->packed && ((st->B >> st->lb) + 1) >= (uint32_t)MSM_MAX_NB1) {   // more than 2^17 buckets
            st->lb = MSM_BIN_LB_WIDE;
            st->lcols = MSM_BIN_LB_WIDE;
        }
    }
    st->nb1 = (st->B >> st->lb) + 1;
    if (st->nb1 >= (uint32_t)MSM_MAX_NB1) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "msm: too many level-1 bins");
    // staged pairs per level-1 workgroup: MSM_L1_CAP of either format; the 4-byte format then needs 60 KB of LDS and two
    // workgroups (32 waves) share a CU, which these latency-bound kernels need more than longer runs
    const uint32_t l1_cap = (uint32_t)MSM_L1_CAP;
    st->l1_scalars = (l1_cap / (uint32_t)st->W) & ~63u;
    if (st->l1_scalars > 1024u) st->l1_scalars = 1024u;
    const size_t max_blk = (count + st->l1_scalars - 1) / st->l1_scalars;
    st->strides.s_bin_offs = up((size_t)st->nb1 * max_blk + 1);
    st->strides.s_bin_aux = up((size_t)st->nb1 * max_blk / MSM_SCAN_TILE + 4);
    st->strides.s_bin = up((size_t)st->nb1 + 1);
    st->l2_items = (uint32_t)(m / MSM_L2_TILE + st->nb1);
    st->strides.s_tile_desc = up(st->l2_items);
    st->strides.s_cnt = up((size_t)st->l2_items << st->lcols);
    if ((rc = dev_alloc(c, (void**)&st->bin_offs, NB * st->strides.s_bin_offs * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->bin_aux, NB * st->strides.s_bin_aux * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->bin_start, NB * st->strides.s_bin * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->tile_start, NB * st->strides.s_bin * 4))) return rc;
    if ((rc = dev_alloc(c, &st->tile_desc, NB * st->strides.s_tile_desc * 8))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->cnt2, NB * st->strides.s_cnt * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->pos2, NB * st->strides.s_cnt * 4))) return rc;
    {
        // the attribute belongs to the kernel, not to this state: size it for the worst case of the instantiation (any
        // number of level-1 bins), so that several states -- contexts in flight, thread-ranks with unequal SRS slices --
        // cannot shrink each other's cap
        const int lds = (int)(((3 * (uint32_t)MSM_MAX_NB1 + 3) & ~3u) * 4 + MSM_L1_CAP * (st->packed ? 4 : 8));
        const void* f = st->packed ? (const void*)msm_pick_bin_scatter<C, PairPacked>(st->dig)
                                   : (const void*)msm_pick_bin_scatter<C, PairWide>(st->dig);
        ZKT_HIP(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    static_assert(MsmState::SLOTS == 11, "per-slot arrays are sized for 11 slots");
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        if ((rc = dev_alloc(c, (void**)&st->offsets[i], (((size_t)st->nb1 << st->lb) + 2) * 4))) return rc;
        if ((rc = dev_alloc(c, (void**)&st->heavy[i], ((size_t)st->B + 2) * 4))) return rc;
        if ((rc = dev_alloc(c, (void**)&st->params[i], 16))) return rc;
    }
    {
        int blocks_per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) cus = prop.multiProcessorCount;
        if (const char* e = exp_env("ZKT_MSM_ACC_LDS")) {   // experiment: cap the resident workgroups through dynamic LDS
            const int kb = atoi(e);
            if (kb < 0 || kb > 160) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "ZKT_MSM_ACC_LDS: 0 .. 160 (KiB)");
            st->acc_lds = (size_t)kb * 1024;
            ZKT_HIP(c, hipFuncSetAttribute((const void*)k_msm_accumulate<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->acc_lds));
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_msm_accumulate<C>, 256, st->acc_lds) == hipSuccess &&
            blocks_per_cu > 0 && cus > 0)
            st->acc_threads = (size_t)blocks_per_cu * cus * 256;
        // Twice as many chunks as the chip keeps resident: the bucket reduction of the previous MSM runs on the side stream
        // while this accumulation starts, and every one of its workgroups holds the registers of an accumulation
        // workgroup for 100-200 us.  With exactly one wave-front of chunks the workgroups that could not start on time
        // ended a whole accumulation late (+8 % on the kernel); with two, the dispatcher gives the delayed CUs fewer of
        // the second half.  (ZKT_MSM_OVER = 1 .. 8 overrides the factor for experiments.)
        int over = 2;
        if (const char* e = exp_env("ZKT_MSM_OVER")) {
            const int f = atoi(e);
            if (f >= 1 && f <= 8) over = f;
        }
        st->acc_threads *= (size_t)over;
    }
    // chunk >= ceil(pairs / acc_threads) pairs per thread (k_msm_scan_aux), so an MSM never cuts its pairs into more than
    // acc_threads chunks (nor more than it has pairs): that bounds the piece array of every slot
    size_t max_chunks = std::min(m, st->acc_threads) + 1;
    for (int i = 0; i < MsmState::SLOTS; ++i)
        if ((rc = dev_alloc(c, &st->pieces[i], (max_chunks + st->B + 2) * sizeof(XyzzRaw<Q>)))) return rc;
    st->strides.s_chunk = up(max_chunks + 2);
    if ((rc = dev_alloc(c, (void**)&st->chunk_bucket, NB * st->strides.s_chunk * 4))) return rc;
    ZKT_HIP(c, hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking));
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        if ((rc = dev_alloc(c, &st->buckets[i], ((size_t)st->B + 1) * sizeof(Xyzz<Q>)))) return rc;
        {   // 2^q1 row sums + 2^q2 column sums of the bucket matrix (q1 + q2 = c - 1)
            const int q = cb - 1, q2 = q / 2, q1 = q - q2;
            if ((rc = dev_alloc(c, &st->rowcol[i], (((size_t)1 << q1) + ((size_t)1 << q2)) * sizeof(Xyzz<Q>)))) return rc;
        }
        ZKT_HIP(c, hipHostMalloc(&st->host_result[i], (size_t)(MSM_MAX_Y + 1) * MSM_R2_BLOCKS * sizeof(Xyzz<Q>), hipHostMallocMapped));
        ZKT_HIP(c, hipHostGetDevicePointer(&st->host_result_dev[i], st->host_result[i], 0));
        ZKT_HIP(c, hipEventCreateWithFlags(&st->ev_main[i], hipEventDisableTiming));
        ZKT_HIP(c, hipEventCreateWithFlags(&st->ev_done[i], hipEventDisableTiming));
    }
    st->defer_tails = count <= MSM_DEFER_MAX;
    if (const char* e = exp_env("ZKT_MSM_DEFER")) st->defer_tails = atoi(e) != 0;
    c->msm = st;
    ++c->msm_epoch;
    ++c->srs_generation;
    return ZKT_OK;
}

template <class C>
static int table_finish(zkt_ctx* c, void* table, size_t count) {
    using Q = typename C::Fq;
    MsmState& st = *c->msm;
    unsigned blocks = (unsigned)((count + 127) / 128);
    hipLaunchKernelGGL(k_srs_windows<C>, dim3(blocks), dim3(128), 0, c->stream, (Affine<Q>*)table, count, st.win);
    ZKT_HIP(c, hipGetLastError());
    const size_t total = (size_t)st.W * count;
    hipLaunchKernelGGL(k_srs_to_fx<C>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       (Affine<Q>*)table, total);
    ZKT_HIP(c, hipGetLastError());
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}
template <class C>
static int srs_finish(zkt_ctx* c) {
    return table_finish<C>(c, c->msm->table, c->msm->count);
}
void msm_release(zkt_ctx* c);
int msm_fork(zkt_ctx* child, const zkt_ctx* parent) {
    msm_release(child);
    if (!parent->msm) return ZKT_OK;
    int rc = parent->curve == ZKT_CURVE_BN254 ? msm_setup<Bn254Curve>(child, parent->msm->count, parent->msm.get())
                                               : msm_setup<Bls381Curve>(child, parent->msm->count, parent->msm.get());
    if (rc) return rc;
    child->srs_generation = parent->srs_generation;
    return ZKT_OK;
}
int msm_table_finish(zkt_ctx* c, void* table, size_t count) {
    if (c->curve == ZKT_CURVE_BN254) return table_finish<Bn254Curve>(c, table, count);
    return table_finish<Bls381Curve>(c, table, count);
}

void msm_release(zkt_ctx* c) {
    if (!c->msm) return;
    MsmState& st = *c->msm;
    (void)hipStreamSynchronize(c->stream);
    if (st.side) (void)hipStreamSynchronize(st.side);
    void* ptrs[] = {st.table_borrowed ? nullptr : st.table, st.table2_borrowed ? nullptr : st.table2, st.vals2, st.pairs,
                    st.bin_offs, st.bin_aux, st.bin_start, st.tile_start, st.cnt2, st.pos2, st.chunk_bucket, st.tile_desc};
    for (void* p : ptrs) dev_free(c, p);
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        dev_free(c, st.heavy[i]); dev_free(c, st.offsets[i]); dev_free(c, st.pieces[i]); dev_free(c, st.params[i]);
        dev_free(c, st.buckets[i]); dev_free(c, st.rowcol[i]);
    }
    c->msm.reset();
}

template <class C>
static int srs_load_t(zkt_ctx* c, const void* src, size_t count, bool src_on_device, size_t slice_off = 0, size_t total = 0) {
    using Q = typename C::Fq;
    if (count == 0) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "empty SRS");
    if (int rf = refuse_if_forked(c, "loading a key")) return rf;
    msm_release(c);
    int rc = msm_setup<C>(c, count);
    if (rc) return rc;
    c->msm->slice_off = slice_off;
    c->msm->total = total ? total : count;
    ZKT_HIP(c, hipMemcpyAsync(c->msm->table, src, count * sizeof(Affine<Q>),
                              src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    {   // bases R^-1 P_i (see the header): one fixed-scalar multiplication per power, once per key
        using R = typename C::Fr;
        Fe<R> raw_one = fe_zero<R>();
        raw_one.v[0] = 1u;
        const Fe<R> rinv = fe_from_mont<R>(raw_one);   // R^-1 mod r as an integer
        hipLaunchKernelGGL((k_srs_scale<C, false>), dim3((unsigned)((count + 127) / 128)), dim3(128), 0, c->stream,
                           (const Affine<Q>*)c->msm->table, (Affine<Q>*)c->msm->table, count, rinv);
        ZKT_HIP(c, hipGetLastError());
    }
    return srs_finish<C>(c);
}

template <class C>
static Affine<typename C::Fq> generator_mont();
template <>
Affine<Bn254Fq> generator_mont<Bn254Curve>() {
    Affine<Bn254Fq> g;
    g.x = fe_from_u32<Bn254Fq>(1);
    g.y = fe_from_u32<Bn254Fq>(2);
    return g;
}
template <>
Affine<Bls381Fq> generator_mont<Bls381Curve>() {
    // ark-bls12-381 G1_GENERATOR_X / _Y (canonical limbs), converted to Montgomery form
    static const uint32_t gx[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                    0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    static const uint32_t gy[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                    0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    Affine<Bls381Fq> g;
    for (int i = 0; i < 12; ++i) {
        g.x.v[i] = gx[i];
        g.y.v[i] = gy[i];
    }
    g.x = fe_to_mont<Bls381Fq>(g.x);
    g.y = fe_to_mont<Bls381Fq>(g.y);
    return g;
}

template <class C>
static int srs_generate_t(zkt_ctx* c, const uint64_t* tau4, size_t count, size_t slice_off = 0, size_t total = 0) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    if (count == 0) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "empty SRS");
    if (int rf = refuse_if_forked(c, "loading a key")) return rf;
    msm_release(c);
    int rc = msm_setup<C>(c, count);
    if (rc) return rc;
    c->msm->slice_off = slice_off;
    c->msm->total = total ? total : count;
    Fe<R> tau;
    memcpy(tau.v, tau4, 32);
    tau = fe_to_mont<R>(tau);
    unsigned blocks = (unsigned)((count + 127) / 128);
    hipLaunchKernelGGL(k_srs_generate<C>, dim3(blocks), dim3(128), 0, c->stream, (Affine<Q>*)c->msm->table, count, tau,
                       generator_mont<C>(), slice_off);
    ZKT_HIP(c, hipGetLastError());
    return srs_finish<C>(c);
}

// The bucket fold and reduction of every slot in st.tail_wait as ONE sequence of four launches (blockIdx.y = slot), behind
// the accumulation of the last of them.
template <class C, bool TI>
static int msm_launch_tails_t(zkt_ctx* c) {
    MsmState& st = *c->msm;
    const int k = st.n_tail_wait;
    if (k == 0) return ZKT_OK;
    st.n_tail_wait = 0;
    MsmTailBatch tb{};
    for (int j = 0; j < MSM_TAIL_BATCH; ++j) {
        const int slot = st.tail_wait[j < k ? j : 0];
        tb.offsets[j] = st.offsets[slot];
        tb.params[j] = st.params[slot];
        tb.pieces[j] = st.pieces[slot];
        tb.buckets[j] = st.buckets[slot];
        tb.heavy[j] = st.heavy[slot];
        tb.rowcol[j] = st.rowcol[slot];
        tb.partials[j] = st.host_result_dev[slot];
    }
    const unsigned ky = (unsigned)k;
    ZKT_HIP(c, hipStreamWaitEvent(st.side, st.ev_main[st.tail_wait[k - 1]], 0));
    {
    ProfScope prof_fold(c, "msm_fold", st.side, (uint64_t)k);
    if (st.defer_tails) {   // small key: sixteen lanes per bucket, no crowded-bucket pass
        hipLaunchKernelGGL((k_msm_bucket_sum_lanes<C, TI>), dim3((16 * (st.B + 1) + 255) / 256, ky), dim3(256), 0, st.side, st.B, tb);
        ZKT_HIP(c, hipGetLastError());
    } else {
        hipLaunchKernelGGL((k_msm_bucket_sum<C, TI>), dim3((st.B + 1 + 255) / 256, ky), dim3(256), 0, st.side, st.B, tb);
        ZKT_HIP(c, hipGetLastError());
        hipLaunchKernelGGL((k_msm_heavy<C, TI>), dim3(MSM_HEAVY_BLOCKS, ky), dim3(256), 0, st.side, tb);
        ZKT_HIP(c, hipGetLastError());
    }
    }
    {
    ProfScope prof_tail(c, "msm_tail", st.side, (uint64_t)k);
    const uint32_t q = (uint32_t)(st.c - 1), q2 = q / 2, q1 = q - q2;   // 2^q buckets below B = 2^q1 rows x 2^q2 columns
    const uint32_t sums = (1u << q1) + (1u << q2);
    hipLaunchKernelGGL((k_msm_rowcol<C, TI>), dim3((sums + 3) / 4, ky), dim3(256), 0, st.side, q1, q2, tb);
    ZKT_HIP(c, hipGetLastError());
    hipLaunchKernelGGL((k_msm_weighted_rows<C, TI>), dim3(q + 1, ky), dim3(64), 0, st.side, q1, q2, st.B, tb);
    ZKT_HIP(c, hipGetLastError());
    }
    // the ny + 1 row sums are written straight into pinned host memory (16 posted writes of 128 B; a copy engine took
    // ~90 us for them): the host finishes the reduction (msm_host_finish) once ev_done has fired
    for (int j = 0; j < k; ++j) ZKT_HIP(c, hipEventRecord(st.ev_done[st.tail_wait[j]], st.side));
    return ZKT_OK;
}

// Inlined products in the tail's additions pay where the tail is the critical path (small keys); on larger keys the tail runs
// beside the next accumulation and its longer code costs more than its shorter chains give (A/B in docs/EXPERIMENTS.md).
template <class C>
static int msm_launch_tails(zkt_ctx* c) {
    // inlined products in the bucket reduction's additions: the deferred regime, and every key up to 2^18 powers (r05: n = 2^17
    // +3.5 % per proof, BLS12-381 +7 %; 2^18 the main stream's idle time halves; 2^19 and 2^20 nothing)
    bool inl = c->msm->defer_tails || c->msm->count <= MSM_TAIL_INL_MAX;
    if (const char* e = exp_env("ZKT_MSM_TAIL_INL")) inl = atoi(e) != 0;
    return inl ? msm_launch_tails_t<C, true>(c) : msm_launch_tails_t<C, false>(c);
}

// Enqueues k <= MSM_BATCH MSMs over the same table as ONE batch: every grouping kernel and the accumulation go out once,
// with blockIdx.y = MSM (the commitments of a prover round are independent: prove.rs:133-135,306-308 -- three launches of a
// latency-bound kernel cost three ramps and tails, one launch of three times the blocks costs one).  The partial sums of MSM j
// land in st.host_result[slots[j]] (pinned), see msm_host_finish.  k = 1 is the single MSM.
template <class C>
static int msm_enqueue_batch(zkt_ctx* c, int k, const void* const* d_scalars, const size_t* ns, const size_t* base_offs, int mont,
                             const int* slots, const int* tbls) {
    MsmState& st = *c->msm;
    if (k < 1 || k > MSM_BATCH) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "msm batch size");
    ++c->msm_epoch;   // slot buffers change hands: anything issued ahead of time that relied on them is stale
    for (int j = 0; j < k; ++j)    // a slot that is enqueued again before its deferred tail went out: issue the tails now
        for (int w = 0; w < st.n_tail_wait; ++w)
            if (st.tail_wait[w] == slots[j])
                if (int rc0 = msm_launch_tails<C>(c)) return rc0;
    if (st.n_tail_wait + k > MSM_TAIL_BATCH)
        if (int rc0 = msm_launch_tails<C>(c)) return rc0;
    // tbls[j] = 1: the Lagrange-prefix table of lagrange.hip (count2 bases) instead of the key's powers
    MsmBatch bt = st.strides;
    bool all_lag = true;
    for (int j = 0; j < k; ++j) all_lag = all_lag && tbls[j] != 0;
    size_t n_max = 0;
    for (int j = 0; j < MSM_BATCH; ++j) {
        const int jj = j < k ? j : 0;
        const int slot = slots[jj];
        // the slot's buffers may still be read by the previous MSM that used this slot (side stream)
        if (j < k && st.pending[slot]) ZKT_HIP(c, hipStreamWaitEvent(c->stream, st.ev_done[slot], 0));
        bt.scalars[j] = d_scalars[jj];
        bt.n[j] = ns[jj];
        bt.base_off[j] = base_offs[jj];
        bt.table[j] = tbls[jj] ? st.table2 : st.table;
        bt.tcount[j] = tbls[jj] ? st.count2 : st.count;
        bt.heavy[j] = st.heavy[slot];
        bt.params[j] = st.params[slot];
        bt.offsets[j] = st.offsets[slot];
        bt.pieces[j] = st.pieces[slot];
        if (j < k) n_max = std::max(n_max, ns[jj]);
    }
    const size_t n = n_max;
    const uint32_t m = (uint32_t)((size_t)st.W * n);
    const unsigned ky = (unsigned)k;
    // (the chunk itself is computed on the device from the pairs that really exist: k_msm_scan_aux -> params[slot])
    {
    // (Lagrange-basis commitments are timed apart: a batch that mixes both counts as dense)
    ProfScope prof_all(c, all_lag ? "msm_lag_main" : "msm_main", nullptr, (uint64_t)k);
    {
        const uint32_t S = st.l1_scalars;
        const unsigned nblk = (unsigned)((n + S - 1) / S);
        const uint32_t total = st.nb1 * nblk, ntiles = (total + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE;
        {
            auto kc = msm_pick_bin_count<C>(st.dig);
            hipLaunchKernelGGL(kc, dim3(nblk, ky), dim3(1024), (size_t)st.nb1 * 4, c->stream, bt, mont, st.win, S, st.nb1, st.lb,
                               st.bin_offs);
        }
        hipLaunchKernelGGL(k_msm_scan_tiles, dim3(ntiles, ky), dim3(1024), 0, c->stream, st.bin_offs, total, st.bin_aux, bt);
        hipLaunchKernelGGL(k_msm_scan_aux, dim3(1, ky), dim3(1024), 0, c->stream, st.bin_offs, st.bin_aux, ntiles, nblk, st.nb1,
                           st.bin_start, st.tile_start, (uint2*)st.tile_desc, (uint32_t)st.acc_threads, st.B,
                           st.defer_tails ? 1u : 0u, bt);
        ZKT_HIP(c, hipGetLastError());
        const size_t lds_scatter = (size_t)((3 * st.nb1 + 3) & ~3u) * 4 + (size_t)MSM_L1_CAP * (st.packed ? 4 : 8);
        const uint32_t items = (uint32_t)(m / MSM_L2_TILE + st.nb1);
        if (st.packed) {
            auto ks = msm_pick_bin_scatter<C, PairPacked>(st.dig);
            hipLaunchKernelGGL(ks, dim3(nblk, ky), dim3(1024), lds_scatter, c->stream, bt, mont, st.win, S,
                               st.nb1, st.lb, st.bin_offs, st.bin_aux, (uint32_t*)st.pairs);
            ZKT_HIP(c, hipGetLastError());
            hipLaunchKernelGGL((k_msm_l2_count<PairPacked, 8>), dim3(items, ky), dim3(256), 0, c->stream, (const uint32_t*)st.pairs,
                               st.nb1, st.lb, st.tile_start, (const uint2*)st.tile_desc, st.cnt2, bt);
        } else {
            auto ks = msm_pick_bin_scatter<C, PairWide>(st.dig);
            hipLaunchKernelGGL(ks, dim3(nblk, ky), dim3(1024), lds_scatter, c->stream, bt, mont, st.win, S,
                               st.nb1, st.lb, st.bin_offs, st.bin_aux, (uint2*)st.pairs);
            ZKT_HIP(c, hipGetLastError());
            auto kc2 = st.lcols == 8 ? k_msm_l2_count<PairWide, 8> : k_msm_l2_count<PairWide, 10>;
            hipLaunchKernelGGL(kc2, dim3(items, ky), dim3(256), 0, c->stream, (const uint2*)st.pairs, st.nb1,
                               st.lb, st.tile_start, (const uint2*)st.tile_desc, st.cnt2, bt);
        }
        {
            auto ksc = st.lcols == 8 ? k_msm_l2_scan<8> : k_msm_l2_scan<10>;
            hipLaunchKernelGGL(ksc, dim3(st.nb1, ky), dim3(256), 0, c->stream, st.cnt2, st.pos2, st.bin_start,
                               st.tile_start, st.B, st.lb, st.chunk_bucket, bt);
        }
        if (st.packed) {
            hipLaunchKernelGGL((k_msm_l2_scatter<PairPacked, 8>), dim3(items, ky), dim3(MSM_L2S_THREADS), 0, c->stream, (const uint32_t*)st.pairs,
                               st.nb1, st.lb, st.tile_start, (const uint2*)st.tile_desc, st.cnt2, st.pos2, st.vals2, bt);
        } else {
            auto kss = st.lcols == 8 ? k_msm_l2_scatter<PairWide, 8> : k_msm_l2_scatter<PairWide, 10>;
            hipLaunchKernelGGL(kss, dim3(items, ky), dim3(MSM_L2S_THREADS), 0, c->stream, (const uint2*)st.pairs, st.nb1,
                               st.lb, st.tile_start, (const uint2*)st.tile_desc, st.cnt2, st.pos2, st.vals2, bt);
        }
        ZKT_HIP(c, hipGetLastError());
    }
    // Accumulation and tail, one MSM after the other even when the grouping was batched: the bucket reduction of MSM j
    // (side stream) then runs beside the accumulation of MSM j + 1, which absorbs it (two wave-fronts of chunks); issued
    // together behind the batch the three tails ran beside the transforms that follow a round and slowed them by 8-10 %.
    for (int j = 0; j < k; ++j) {
        MsmBatch one = bt;
        one.offsets[0] = bt.offsets[j];
        one.pieces[0] = bt.pieces[j];
        one.params[0] = bt.params[j];
        one.table[0] = bt.table[j];
        {
            ProfScope prof_acc(c, tbls[j] ? "msm_lag_accumulate" : "msm_accumulate");
            // as many threads as an MSM can have chunks (threads past the last chunk leave at once)
            const uint32_t max_chunks = (uint32_t)std::min((size_t)st.W * ns[j], st.acc_threads);
            hipLaunchKernelGGL(k_msm_accumulate<C>, dim3((max_chunks + 255) / 256, 1), dim3(256), st.acc_lds, c->stream,
                               st.vals2 + (size_t)j * st.strides.s_vals, st.B, st.chunk_bucket + (size_t)j * st.strides.s_chunk, one);
            ZKT_HIP(c, hipGetLastError());
        }
        // ---- tail on the side stream: the bucket fold (latency bound: one wave per SIMD, a few dependent additions) and
        // the bucket reduction overlap whatever the main stream does next; everything they read is the slot's own ----
        ZKT_HIP(c, hipEventRecord(st.ev_main[slots[j]], c->stream));
        st.tail_wait[st.n_tail_wait++] = slots[j];
        st.pending[slots[j]] = true;
        if (st.defer_tails && st.n_tail_wait < MSM_TAIL_BATCH) continue;   // msm_flush_tails, once per round
        if (int rc = msm_launch_tails<C>(c)) return rc;
    }
    }   // "msm_main": grouping and accumulations of the batch (the tails' launches on the side stream cost it nothing)
    return ZKT_OK;
}

template <class C>
static int msm_enqueue(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, int slot = 0, int tbl = 0) {
    return msm_enqueue_batch<C>(c, 1, &d_scalars, &n, &base_off, mont, &slot, &tbl);
}

// S = sum_y 2^y V_y : V_y = the sum of the buckets below B whose index has bit y (y < c - 1), V_(c-1) = the top bucket.
// ~50 curve operations on 64-bit limbs.
template <class Q>
static Xyzz<Q> msm_host_finish(const MsmState& st, int slot) {
    const int ny = msm_rows(st.c);
    int exps[MSM_MAX_Y + 1];
    for (int y = 0; y <= ny; ++y) exps[y] = y;
    return hostec::weighted_row_sum<Q>((const Xyzz<Q>*)st.host_result[slot], ny + 1, MSM_R2_BLOCKS, exps);
}

// waits for the MSM in `slot` and normalises its result on the host (one inversion; the affine
// coordinates are needed there for the Fiat-Shamir transcript anyway)
template <class C>
static int msm_collect(zkt_ctx* c, int slot, Affine<typename C::Fq>* out) {
    using Q = typename C::Fq;
    MsmState& st = *c->msm;
    if (int rc0 = msm_launch_tails<C>(c)) return rc0;
    ZKT_HIP(c, hipEventSynchronize(st.ev_done[slot]));
    st.pending[slot] = false;
    *out = xyzz_to_affine_host<Q>(msm_host_finish<Q>(st, slot));
    return ZKT_OK;
}

// Collects a batch of commitments whose points are sharded across the GPUs of the communicator by index range
// (SURVEY.md 8e "MSM - one tiny exchange"): every rank contributes the XYZZ partial sum of its slice (the identity where
// the polynomial does not reach into it, have[j] = false: no MSM was started in that slot), ONE all-gather moves the
// k x world partial sums as raw bytes (a collective cannot reduce curve points), and each rank adds them in rank order and
// normalises.  Every rank obtains the same k affine points, bit for bit.
template <class C>
static int msm_collect_sharded(zkt_ctx* c, const int* slots, const bool* have, int k, Affine<typename C::Fq>* out) {
    using Q = typename C::Fq;
    MsmState& st = *c->msm;
    const int world = c->comm.vt.world;
    std::vector<Xyzz<Q>> send(k), recv((size_t)k * world);
    if (int rc0 = msm_launch_tails<C>(c)) return rc0;
    for (int j = 0; j < k; ++j) {
        if (have[j]) {
            ZKT_HIP(c, hipEventSynchronize(st.ev_done[slots[j]]));
            st.pending[slots[j]] = false;
            send[j] = msm_host_finish<Q>(st, slots[j]);
        } else {
            send[j] = xyzz_identity<Q>();
        }
    }
    int rc = comm_all_gather_host(c, send.data(), recv.data(), (size_t)k * sizeof(Xyzz<Q>));
    if (rc) return rc;
    for (int j = 0; j < k; ++j) {
        Xyzz<Q> acc = xyzz_identity<Q>();
        for (int r = 0; r < world; ++r) acc = xyzz_add<Q>(acc, recv[(size_t)r * k + j]);
        out[j] = xyzz_to_affine_host<Q>(acc);
    }
    return ZKT_OK;
}

template <class C>
static int msm_run_t(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, uint64_t* out_xy,
                     int* out_inf) {
    using Q = typename C::Fq;
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    MsmState& st = *c->msm;
    if (base_off > st.count || n > st.count - base_off)
        return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
    Affine<Q> res;
    if (n == 0) {
        res.x = fe_zero<Q>();
        res.y = fe_zero<Q>();
    } else {
        int rc = msm_enqueue<C>(c, d_scalars, n, base_off, mont, 0);
        if (rc) return rc;
        if ((rc = msm_collect<C>(c, 0, &res))) return rc;
    }
    memcpy(out_xy, res.x.v, Q::N * 4);
    memcpy(out_xy + Q::N / 2, res.y.v, Q::N * 4);
    if (out_inf) *out_inf = aff_is_inf<Q>(res) ? 1 : 0;
    return ZKT_OK;
}

// prover-facing batch form: begin up to MsmState::SLOTS commitments, then collect them
int msm_begin(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, int slot, int tbl) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (slot < 0 || slot >= MsmState::SLOTS) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "bad MSM slot");
    if (tbl && !c->msm->table2) return set_err(c, ZKT_ERR_NOT_LOADED, "no Lagrange-basis table built");
    const size_t cnt = tbl ? c->msm->count2 : c->msm->count;
    if (n == 0 || base_off > cnt || n > cnt - base_off)
        return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
    if (c->curve == ZKT_CURVE_BN254) return msm_enqueue<Bn254Curve>(c, d_scalars, n, base_off, mont, slot, tbl);
    return msm_enqueue<Bls381Curve>(c, d_scalars, n, base_off, mont, slot, tbl);
}
// k commitments over the key's powers as one batch (msm_enqueue_batch); slots: k distinct slots
int msm_begin_batch(zkt_ctx* c, int k, const void* const* d_scalars, const size_t* ns, int mont, const int* slots, const int* tbls) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (k < 1 || k > MSM_BATCH) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "msm batch size");
    size_t offs[MSM_BATCH] = {};
    for (int j = 0; j < k; ++j) {
        if (slots[j] < 0 || slots[j] >= MsmState::SLOTS) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "bad MSM slot");
        for (int i = 0; i < j; ++i)
            if (slots[i] == slots[j]) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "msm batch: slots must differ");
        if (tbls[j] && !c->msm->table2) return set_err(c, ZKT_ERR_NOT_LOADED, "no Lagrange-basis table built");
        if (ns[j] == 0 || ns[j] > (tbls[j] ? c->msm->count2 : c->msm->count))
            return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
    }
    if (c->curve == ZKT_CURVE_BN254) return msm_enqueue_batch<Bn254Curve>(c, k, d_scalars, ns, offs, mont, slots, tbls);
    return msm_enqueue_batch<Bls381Curve>(c, k, d_scalars, ns, offs, mont, slots, tbls);
}
bool msm_defers_tails(const zkt_ctx* c) { return c->msm && c->msm->defer_tails; }
// Grouping the commitments of a round as one batch of launches pays between the two regimes: below, the proof is a chain
// of latencies (tails deferred instead); above (n = 2^20: 190 MB of grouped indices per batch) the accumulation finds the
// indices of its own MSM evicted from the last-level cache by its neighbours' and runs 4-5 % longer, more than the grouping saves.
bool msm_batches_grouping(const zkt_ctx* c) {
    if (!c->msm) return false;
    // measured: +5 % at 2^14, +4 % at 2^16, -2 % at 2^17, +3 % at 2^18 (with the inlined tail additions; BLS12-381 +2 %),
    // nothing at 2^19, -1 % at 2^20
    const size_t n = c->msm->count;
    if (const char* e = exp_env("ZKT_MSM_BATCH_MAX_LOG")) return n <= ((size_t)1 << atoi(e)) + 64;
    return n <= MSM_DEFER_MAX || (n > ((size_t)1 << 17) + 64 && n <= ((size_t)1 << 18) + 64);
}
int msm_flush_tails(zkt_ctx* c) {
    if (!c->msm) return ZKT_OK;
    if (c->curve == ZKT_CURVE_BN254) return msm_launch_tails<Bn254Curve>(c);
    return msm_launch_tails<Bls381Curve>(c);
}
int msm_end(zkt_ctx* c, int slot, uint64_t* out_xy) {
    if (c->curve == ZKT_CURVE_BN254) {
        Affine<Bn254Fq> a;
        int rc = msm_collect<Bn254Curve>(c, slot, &a);
        if (rc) return rc;
        memcpy(out_xy, a.x.v, 32);
        memcpy(out_xy + 4, a.y.v, 32);
        return ZKT_OK;
    }
    Affine<Bls381Fq> a;
    int rc = msm_collect<Bls381Curve>(c, slot, &a);
    if (rc) return rc;
    memcpy(out_xy, a.x.v, 48);
    memcpy(out_xy + 6, a.y.v, 48);
    return ZKT_OK;
}

int msm_end_sharded(zkt_ctx* c, const int* slots, const bool* have, int k, uint64_t* out_xy /* k x 12 words */) {
    if (c->curve == ZKT_CURVE_BN254) {
        std::vector<Affine<Bn254Fq>> a(k);
        int rc = msm_collect_sharded<Bn254Curve>(c, slots, have, k, a.data());
        if (rc) return rc;
        for (int j = 0; j < k; ++j) {
            memcpy(out_xy + 12 * j, a[j].x.v, 32);
            memcpy(out_xy + 12 * j + 4, a[j].y.v, 32);
        }
        return ZKT_OK;
    }
    std::vector<Affine<Bls381Fq>> a(k);
    int rc = msm_collect_sharded<Bls381Curve>(c, slots, have, k, a.data());
    if (rc) return rc;
    for (int j = 0; j < k; ++j) {
        memcpy(out_xy + 12 * j, a[j].x.v, 48);
        memcpy(out_xy + 12 * j + 6, a[j].y.v, 48);
    }
    return ZKT_OK;
}
void msm_slice(zkt_ctx* c, size_t* off, size_t* count, size_t* total) {
    *off = c->msm ? c->msm->slice_off : 0;
    *count = c->msm ? c->msm->count : 0;
    *total = c->msm ? c->msm->total : 0;
}

int msm_g1_dev(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, uint64_t* out_xy, int* out_inf) {
    if (c->curve == ZKT_CURVE_BN254) return msm_run_t<Bn254Curve>(c, d_scalars, n, base_off, mont, out_xy, out_inf);
    return msm_run_t<Bls381Curve>(c, d_scalars, n, base_off, mont, out_xy, out_inf);
}
int msm_enqueue_only(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (base_off > c->msm->count || n > c->msm->count - base_off || n == 0)
        return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients");
    if (c->curve == ZKT_CURVE_BN254) return msm_enqueue<Bn254Curve>(c, d_scalars, n, base_off, mont);
    return msm_enqueue<Bls381Curve>(c, d_scalars, n, base_off, mont);
}
int srs_load(zkt_ctx* c, const void* src, size_t count, bool on_device, size_t slice_off = 0, size_t total = 0) {
    if (c->curve == ZKT_CURVE_BN254) return srs_load_t<Bn254Curve>(c, src, count, on_device, slice_off, total);
    return srs_load_t<Bls381Curve>(c, src, count, on_device, slice_off, total);
}
int srs_generate(zkt_ctx* c, const uint64_t* tau4, size_t count, size_t slice_off = 0, size_t total = 0) {
    if (c->curve == ZKT_CURVE_BN254) return srs_generate_t<Bn254Curve>(c, tau4, count, slice_off, total);
    return srs_generate_t<Bls381Curve>(c, tau4, count, slice_off, total);
}
template <class C>
static int srs_download_t(zkt_ctx* c, size_t offset, size_t count, uint64_t* out) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    // the table holds R^-1 P_i in R' form: hand back the caller's own powers, P_i = [R] (R^-1 P_i), in arkworks' form
    // (test / bench aid; a fixed-scalar multiplication per point on the device)
    if (count == 0) return ZKT_OK;
    void* tmp = nullptr;
    int rc = dev_alloc(c, &tmp, count * sizeof(Affine<Q>));
    if (rc) return rc;
    const Fe<R> r_canon = fe_one<R>();   // the Montgomery form of one = R mod r, read as an integer
    hipLaunchKernelGGL((k_srs_scale<C, true>), dim3((unsigned)((count + 127) / 128)), dim3(128), 0, c->stream,
                       (const Affine<Q>*)c->msm->table + offset, (Affine<Q>*)tmp, count, r_canon);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, count * sizeof(Affine<Q>), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, tmp);
    return e == hipSuccess ? ZKT_OK : hip_fail(c, e, "srs_download");
}

int srs_download(zkt_ctx* c, size_t offset, size_t count, uint64_t* out) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded");
    if (offset > c->msm->count || count > c->msm->count - offset) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "range");
    if (c->curve == ZKT_CURVE_BN254) return srs_download_t<Bn254Curve>(c, offset, count, out);
    return srs_download_t<Bls381Curve>(c, offset, count, out);
}
void msm_info(zkt_ctx* c, int* cbits, int* windows, size_t* count) {
    if (!c->msm) {
        *cbits = 0; *windows = 0; *count = 0;
        return;
    }
    *cbits = c->msm->c;
    *windows = c->msm->W;
    *count = c->msm->count;
}

}  // namespace zkt

using namespace zkt;

extern "C" {

int zkt_srs_load(zkt_ctx* c, const uint64_t* g1_xy_mont, size_t count) {
    if (!c || !g1_xy_mont) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return srs_load(c, g1_xy_mont, count, false);
}
int zkt_srs_load_dev(zkt_ctx* c, const void* d_g1_xy_mont, size_t count) {
    if (!c || !d_g1_xy_mont) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return srs_load(c, d_g1_xy_mont, count, true);
}
int zkt_srs_generate(zkt_ctx* c, const uint64_t* tau_canonical4, size_t count) {
    if (!c || !tau_canonical4) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return srs_generate(c, tau_canonical4, count);
}
int zkt_srs_load_slice(zkt_ctx* c, const uint64_t* g1_xy_mont_slice, size_t offset, size_t count, size_t total) {
    if (!c || !g1_xy_mont_slice) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (offset > total || count > total - offset) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "slice outside the key");
    (void)hipSetDevice(c->device);
    return srs_load(c, g1_xy_mont_slice, count, false, offset, total);
}
int zkt_srs_generate_slice(zkt_ctx* c, const uint64_t* tau_canonical4, size_t offset, size_t count, size_t total) {
    if (!c || !tau_canonical4) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (offset > total || count > total - offset) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "slice outside the key");
    (void)hipSetDevice(c->device);
    return srs_generate(c, tau_canonical4, count, offset, total);
}
int zkt_srs_download(zkt_ctx* c, size_t offset, size_t count, uint64_t* out_xy_mont) {
    if (!c || !out_xy_mont) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return srs_download(c, offset, count, out_xy_mont);
}

int zkt_msm_g1_dev(zkt_ctx* c, const void* d_scalars, size_t len, size_t base_offset, int scalars_montgomery,
                   void* out_xy_mont_host) {
    if (!c || (!d_scalars && len) || !out_xy_mont_host) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return msm_g1_dev(c, d_scalars, len, base_offset, scalars_montgomery, (uint64_t*)out_xy_mont_host, nullptr);
}

int zkt_msm_g1(zkt_ctx* c, const uint64_t* scalars, size_t len, size_t base_offset, int scalars_montgomery,
               uint64_t* out_xy_mont, int* out_is_infinity) {
    if (!c || (!scalars && len) || !out_xy_mont) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    if (len) {
        int rc = ensure_buffer(c, &c->io_a, &c->io_a_bytes, len * 32);
        if (rc) return rc;
        ZKT_HIP(c, hipMemcpyAsync(c->io_a, scalars, len * 32, hipMemcpyHostToDevice, c->stream));
    }
    return msm_g1_dev(c, c->io_a, len, base_offset, scalars_montgomery, out_xy_mont, out_is_infinity);
}

int zkt_msm_enqueue_dev(zkt_ctx* c, const void* d_scalars, size_t len, size_t base_offset, int scalars_montgomery) {
    if (!c || !d_scalars) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return msm_enqueue_only(c, d_scalars, len, base_offset, scalars_montgomery);
}

int zkt_msm_info(zkt_ctx* c, int* window_bits, int* windows, size_t* srs_count) {
    if (!c || !window_bits || !windows || !srs_count) return ZKT_ERR_INVALID_ARGUMENT;
    msm_info(c, window_bits, windows, srs_count);
    return ZKT_OK;
}

}  // extern "C"
