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
//  * sum_b b * B_b is computed with short per-thread running sums (segments of 8 buckets), masked
//    wave/block tree reductions for the segment weights and a final doubling step.
//  * The single resulting point is normalised (one inversion) on the host, which needs the affine
//    coordinates for the Fiat-Shamir transcript anyway.
#include "ctx.hpp"
#include "ec.hpp"
#include "ecx.hpp"
#include "hostec.hpp"

#include <algorithm>
#include <cstring>

namespace zkt {

constexpr int MSM_CHUNK_MIN = 16;  // grouped pairs per accumulation thread: at least this many; the actual
                                   // chunk is sized so that ONE resident wave-front of threads covers the array
constexpr int MSM_LOG_SEG = 2;
constexpr int MSM_SEG = 1 << MSM_LOG_SEG;   // buckets per running-sum segment
constexpr int MSM_R2_BLOCKS = 8;
constexpr int MSM_MAX_Y = 24;

// Window layout: W windows of width c or c-1 covering exactly lambda+1 bits, so that no window
// (in particular not the top one) is left with only a few significant bits: a 2-bit top window
// would pour n entries into 4 buckets.
struct MsmWindows {
    int W;
    uint8_t width[40];
    uint16_t start[40];
};

constexpr int MSM_HEAVY = 32;       // buckets with more pieces than this are folded by a whole block
constexpr int MSM_HEAVY_BLOCKS = 512;

struct MsmState {
    size_t count = 0;      // bases loaded
    // index-range sharding: this GPU holds powers [slice_off, slice_off + count) of a key of `total`
    size_t slice_off = 0, total = 0;
    int c = 0, W = 0;      // max window bits, windows
    MsmWindows win{};
    uint32_t* heavy[11] = {};   // per slot: [0] = count, [1..] = heavy bucket ids
    uint32_t B = 0;        // buckets = 2^(c-1), ids 1..B
    void* table = nullptr; // Affine[W][count]
    // work buffers (sized for n = count)
    uint32_t* vals2 = nullptr;                     // table indices grouped by bucket
    void* pairs = nullptr;                         // uint2[m]: after the level-1 split
    uint32_t* bin_offs = nullptr;                  // [nb1][blocks] level-1 counts, scanned per 4096-tile
    uint32_t* bin_aux = nullptr;                   // tile totals, scanned; last = number of pairs
    uint32_t *bin_start = nullptr, *tile_start = nullptr;   // nb1 + 1 each: level-2 work list
    uint32_t *cnt2 = nullptr, *pos2 = nullptr;     // [level-2 tiles][256]
    uint32_t* chunk_bucket = nullptr;              // bucket of the first pair of every accumulation chunk
    uint32_t nb1 = 0;                              // level-1 bins
    uint32_t l1_scalars = 0;                       // scalars per level-1 workgroup
    uint32_t l2_items = 0;                         // upper bound of level-2 tiles
    // per slot, because the bucket fold that reads them runs on the side stream while the next MSM is already grouping
    uint32_t* offsets[11] = {};   // B + 2
    void* pieces[11] = {};        // XyzzRaw[max_chunks + B + 2]
    // The latency-bound tail of an MSM (bucket reduction) runs on a side stream so that it overlaps the
    // next MSM's accumulation; each in-flight MSM owns one slot of tail buffers.
    static constexpr int SLOTS = 11;
    void* buckets[SLOTS] = {};      // Xyzz[B + 1]
    void* segA[SLOTS] = {};         // Xyzz[B / SEG]
    void* segT[SLOTS] = {};
    void* host_result[SLOTS] = {};  // pinned: the (rows + 1) x R2_BLOCKS partial sums the host finishes
    void* host_result_dev[SLOTS] = {};  // the same memory as the kernels address it
    size_t acc_threads = 196608;   // resident threads of k_msm_accumulate (occupancy query at setup)
    hipStream_t side = nullptr;
    hipEvent_t ev_main[SLOTS] = {}, ev_done[SLOTS] = {};
    bool pending[SLOTS] = {};
    ~MsmState() {
        for (int i = 0; i < SLOTS; ++i) {
            if (host_result[i]) (void)hipHostFree(host_result[i]);
            if (ev_main[i]) (void)hipEventDestroy(ev_main[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
        }
        if (side) (void)hipStreamDestroy(side);
    }
};

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
    Fe<R> s = fe_from_mont<R>(fe_pow_u64<R>(tau_mont, (uint64_t)(first + i)));
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
constexpr int MSM_BIN_LB = 8;             // level-2 key bits
constexpr int MSM_L1_CAP = 15360;         // pairs staged per level-1 workgroup (120 KB of LDS)
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

template <class C>
__global__ __launch_bounds__(1024) void k_msm_bin_count(const Fe<typename C::Fr>* scalars, size_t n, int mont,
                                                        MsmWindows win, uint32_t per_block, uint32_t nb1,
                                                        uint32_t* counts) {
    using R = typename C::Fr;
    extern __shared__ uint32_t lds[];
    uint32_t* hist = lds;
    for (uint32_t b = threadIdx.x; b < nb1; b += 1024) hist[b] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * per_block + threadIdx.x;
    if (threadIdx.x < per_block && i < n) {
        Fe<R> s = fe_load<R>(scalars + i);
        if (mont) s = fe_from_mont<R>(s);
        msm_for_each_digit<R>(s, win, [&](int, uint32_t d, uint32_t) {
            if (d) atomicAdd(&hist[d >> MSM_BIN_LB], 1u);
        });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb1; b += 1024) counts[(size_t)b * gridDim.x + blockIdx.x] = hist[b];
}

// exclusive scan of the level-1 counts in two small launches: 4096-element tiles scanned in place, then the
// tile totals (aux).  Readers add the two parts themselves: off(i) = counts[i] + aux[i >> 12].
constexpr int MSM_SCAN_TILE = 4096;
__global__ __launch_bounds__(1024) void k_msm_scan_tiles(uint32_t* counts, uint32_t total, uint32_t* aux) {
    __shared__ uint32_t wsum[16];
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
                                                       uint32_t* heavy_count) {
    __shared__ uint32_t wsum[16];
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
}

// LDS: cursor[nb1] | delta[nb1] | stage uint2[MSM_L1_CAP]
template <class C>
__global__ __launch_bounds__(1024) void k_msm_bin_scatter(const Fe<typename C::Fr>* scalars, size_t n, int mont,
                                                          MsmWindows win, uint32_t per_block, size_t count,
                                                          size_t base_off, uint32_t nb1, const uint32_t* offs,
                                                          const uint32_t* aux, uint2* pairs) {
    using R = typename C::Fr;
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wsum[16];
    uint32_t* cursor = lds;
    uint32_t* delta = lds + nb1;
    uint2* stage = (uint2*)(lds + 2 * nb1);
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
        delta[threadIdx.x] = g0 - ex;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * per_block + threadIdx.x;
    if (threadIdx.x < per_block && i < n) {
        Fe<R> s = fe_load<R>(scalars + i);
        if (mont) s = fe_from_mont<R>(s);
        msm_for_each_digit<R>(s, win, [&](int w, uint32_t d, uint32_t neg) {
            if (d) {
                const uint32_t at = atomicAdd(&cursor[d >> MSM_BIN_LB], 1u);
                stage[at] = make_uint2(d, (uint32_t)((size_t)w * count + base_off + i) | (neg << 31));
            }
        });
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < staged; j += 1024) {
        const uint2 kv = stage[j];
        pairs[delta[kv.x >> MSM_BIN_LB] + j] = kv;
    }
}

struct L2Item {
    uint32_t bin, s, e;
    bool valid;
};
ZKT_D L2Item msm_l2_item(uint32_t item, uint32_t nb1, const uint32_t* bin_start, const uint32_t* tile_start) {
    L2Item r;
    r.valid = item < tile_start[nb1];
    r.bin = 0; r.s = 0; r.e = 0;
    if (!r.valid) return r;
    uint32_t lo = 0, hi = nb1 - 1;          // last bin with tile_start[bin] <= item
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= item) lo = mid; else hi = mid - 1;
    }
    r.bin = lo;
    const uint32_t sub = item - tile_start[lo];
    r.s = bin_start[lo] + sub * MSM_L2_TILE;
    const uint32_t be = bin_start[lo + 1];
    r.e = (r.s + MSM_L2_TILE < be) ? r.s + MSM_L2_TILE : be;
    return r;
}

__global__ __launch_bounds__(256) void k_msm_l2_count(const uint2* pairs, uint32_t nb1, const uint32_t* bin_start,
                                                      const uint32_t* tile_start, uint32_t* cnt2) {
    __shared__ uint32_t hist[256];
    const L2Item it = msm_l2_item(blockIdx.x, nb1, bin_start, tile_start);
    if (!it.valid) return;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t p = it.s + threadIdx.x; p < it.e; p += 256) atomicAdd(&hist[pairs[p].x & 255u], 1u);
    __syncthreads();
    cnt2[(size_t)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];
}

// one workgroup per bin: positions of every (tile, bucket) run and offsets[bucket]
__global__ __launch_bounds__(256) void k_msm_l2_scan(const uint32_t* cnt2, uint32_t* pos2, const uint32_t* bin_start,
                                                     const uint32_t* tile_start, uint32_t* offsets, uint32_t B,
                                                     uint32_t chunk, uint32_t* chunk_bucket) {
    __shared__ uint32_t tot[256];
    const uint32_t b = blockIdx.x;
    const uint32_t t0 = tile_start[b], t1 = tile_start[b + 1];
    uint32_t run = 0;
    for (uint32_t t = t0; t < t1; ++t) {
        const size_t at = (size_t)t * 256 + threadIdx.x;
        pos2[at] = run;
        run += cnt2[at];
    }
    tot[threadIdx.x] = run;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (threadIdx.x >= (uint32_t)d) ? tot[threadIdx.x - d] : 0u;
        __syncthreads();
        tot[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t first = bin_start[b] + tot[threadIdx.x] - run;
    const uint32_t key = b * 256u + threadIdx.x;
    offsets[key] = first;   // sized nb1 * 256 + 2; keys above B are empty and repeat the end
    // accumulation chunk t starts at pair t * chunk: tell it which bucket that pair belongs to.  A bucket normally
    // covers a handful of chunks; a crowded one (skewed digits) is written by the whole workgroup.
    __shared__ uint32_t big[256][3];
    __shared__ uint32_t nbig;
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
    if (key >= 1 && key <= B && run) {
        const uint32_t tb = (first + chunk - 1) / chunk;
        const uint32_t te = (uint32_t)(((uint64_t)first + run + chunk - 1) / chunk);   // one past the last chunk start inside
        if (te - tb > 64) {
            const uint32_t at = atomicAdd(&nbig, 1u);
            big[at][0] = key; big[at][1] = tb; big[at][2] = te;
        } else {
            for (uint32_t t = tb; t < te; ++t) chunk_bucket[t] = key;
        }
    }
    __syncthreads();
    for (uint32_t e = 0; e < nbig; ++e)
        for (uint32_t t = big[e][1] + threadIdx.x; t < big[e][2]; t += 256) chunk_bucket[t] = big[e][0];
    for (uint32_t t = t0; t < t1; ++t) pos2[(size_t)t * 256 + threadIdx.x] += first;
}

__global__ __launch_bounds__(256) void k_msm_l2_scatter(const uint2* pairs, uint32_t nb1, const uint32_t* bin_start,
                                                        const uint32_t* tile_start, const uint32_t* cnt2,
                                                        const uint32_t* pos2, uint32_t* vals) {
    __shared__ uint32_t cursor[256], delta[256];
    __shared__ uint32_t sval[MSM_L2_TILE];
    __shared__ uint8_t skey[MSM_L2_TILE];
    const L2Item it = msm_l2_item(blockIdx.x, nb1, bin_start, tile_start);
    if (!it.valid) return;
    const size_t at = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t mine = cnt2[at];
    cursor[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (threadIdx.x >= (uint32_t)d) ? cursor[threadIdx.x - d] : 0u;
        __syncthreads();
        cursor[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t ex = cursor[threadIdx.x] - mine;
    __syncthreads();
    cursor[threadIdx.x] = ex;
    delta[threadIdx.x] = pos2[at] - ex;
    __syncthreads();
    for (uint32_t p = it.s + threadIdx.x; p < it.e; p += 256) {
        const uint2 kv = pairs[p];
        const uint32_t low = kv.x & 255u;
        const uint32_t a = atomicAdd(&cursor[low], 1u);
        sval[a] = kv.y;
        skey[a] = (uint8_t)low;
    }
    __syncthreads();
    const uint32_t cnt = it.e - it.s;
    for (uint32_t j = threadIdx.x; j < cnt; j += 256) vals[delta[skey[j]] + j] = sval[j];
}

// ---------------------------------------------------------------------------------------------
// accumulation: fixed-size chunks over the grouped table indices, one piece per (chunk, bucket).
// Coordinates live in registers as lazily reduced 29-bit limbs (ecx.hpp); the table and the buckets are canonical
// packed words in R' Montgomery form, the pieces are the raw limbs (XyzzRaw).
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256) void k_msm_accumulate(const uint32_t* vals, uint32_t B, uint32_t chunk,
                                                        const uint32_t* offsets, const uint32_t* chunk_bucket,
                                                        const Affine<typename C::Fq>* table,
                                                        XyzzRaw<typename C::Fq>* pieces) {
    using Q = typename C::Fq;
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

template <class C>
__global__ __launch_bounds__(256) void k_msm_bucket_sum(const uint32_t* offsets, uint32_t B, uint32_t chunk,
                                                        const XyzzRaw<typename C::Fq>* pieces,
                                                        Xyzz<typename C::Fq>* buckets, uint32_t* heavy) {
    using Q = typename C::Fq;
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
            for (uint32_t t = t0 + 1; t <= t1; ++t) acc = xx_add<Q>(acc, xx_load_raw<Q>(pieces + (size_t)t + b));
        }
    }
    xx_store<Q>(buckets + b, acc);
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
template <class Q>
ZKT_D XyzzX<Q> block_sum_256(XyzzX<Q> acc, Xyzz<Q>* wsum) {
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        XyzzX<Q> o = xx_shfl_down<Q>(acc, d);
        if ((threadIdx.x & 63) + d >= 64) o = xx_identity<Q>();
        acc = xx_add<Q>(acc, o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) xx_store<Q>(wsum + wv, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = xx_load<Q>(wsum);
        for (int i = 1; i < 4; ++i) acc = xx_add<Q>(acc, xx_load<Q>(wsum + i));
    }
    __syncthreads();
    return acc;
}

// crowded buckets (skewed digit distributions): one block folds all pieces of one bucket
template <class C>
__global__ __launch_bounds__(256) void k_msm_heavy(const uint32_t* offsets, uint32_t chunk,
                                                   const XyzzRaw<typename C::Fq>* pieces,
                                                   Xyzz<typename C::Fq>* buckets, const uint32_t* heavy) {
    using Q = typename C::Fq;
    __shared__ Xyzz<Q> wsum[4];
    const uint32_t nheavy = heavy[0];
    const uint32_t base = offsets[1];
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy[1 + h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t t0 = (s - base) / chunk, t1 = (e - 1 - base) / chunk;
        XyzzX<Q> acc = xx_identity<Q>();
        for (uint32_t t = t0 + threadIdx.x; t <= t1; t += 256) acc = xx_add<Q>(acc, xx_load_raw<Q>(pieces + (size_t)t + b));
        acc = block_sum_256<Q>(acc, wsum);
        if (threadIdx.x == 0) xx_store<Q>(buckets + b, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// bucket reduction  S = sum_{b=0..B} b * bucket[b]
//   = 2^(c-1) * bucket[B] + sum_s A_s + SEG * sum_s s * T_s      (s over segments of SEG buckets)
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256) void k_msm_segments(const Xyzz<typename C::Fq>* buckets, uint32_t nseg,
                                                      Xyzz<typename C::Fq>* segA, Xyzz<typename C::Fq>* segT) {
    using Q = typename C::Fq;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const Xyzz<Q>* a = buckets + (size_t)s * MSM_SEG;
    XyzzX<Q> run = xx_identity<Q>(), acc = xx_identity<Q>();
#pragma unroll 1
    for (int j = MSM_SEG - 1; j >= 1; --j) {
        run = xx_add<Q>(run, xx_load<Q>(a + j));
        acc = xx_add<Q>(acc, run);
    }
    run = xx_add<Q>(run, xx_load<Q>(a));
    xx_store<Q>(segA + s, acc);
    xx_store<Q>(segT + s, run);
}

// y = 0: plain sum of segA ; y = k + 1 (< ny): sum of segT[s] over s with bit k set ; y = ny: the top bucket alone.
// One partial per block, written in arkworks' R form: the host adds them up and applies the weights
// (hostec.hpp weighted_row_sum) -- the remaining ~35 dependent curve operations cost a wavefront 0.6 ms and the host 15 us.
template <class C>
__global__ __launch_bounds__(256) void k_msm_masked_sums(const Xyzz<typename C::Fq>* segA,
                                                         const Xyzz<typename C::Fq>* segT, uint32_t nseg, int ny,
                                                         const Xyzz<typename C::Fq>* top_bucket,
                                                         Xyzz<typename C::Fq>* partials) {
    using Q = typename C::Fq;
    __shared__ Xyzz<Q> wsum[4];
    const int y = blockIdx.y;
    if (y == ny) {
        if (threadIdx.x == 0)
            xx_store_ark<Q>(partials + (size_t)y * gridDim.x + blockIdx.x, blockIdx.x == 0 ? xx_load<Q>(top_bucket) : xx_identity<Q>());
        return;
    }
    const Xyzz<Q>* src = (y == 0) ? segA : segT;
    XyzzX<Q> acc = xx_identity<Q>();
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
        if (y == 0 || ((s >> (y - 1)) & 1u)) acc = xx_add<Q>(acc, xx_load<Q>(src + s));
    }
    acc = block_sum_256<Q>(acc, wsum);
    if (threadIdx.x == 0) xx_store_ark<Q>(partials + (size_t)y * gridDim.x + blockIdx.x, acc);
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
// rows of partial sums an MSM leaves for the host: row 0 (weight 1), one row per bit of the segment index, the top bucket
static int msm_rows(int c) { return 1 + (c - 1 - MSM_LOG_SEG); }

static int floor_log2(size_t x) {
    int l = 0;
    while ((x >> (l + 1)) != 0) ++l;
    return l;
}

template <class C>
static int msm_setup(zkt_ctx* c, size_t count) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    auto st = std::make_shared<MsmState>();
    st->count = count;
    int lg = floor_log2(count ? count : 1);
    int cb = lg - 2;
    if (cb < 8) cb = 8;
    if (cb > 18) cb = 18;
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
    }
    st->B = 1u << (cb - 1);
    if ((uint64_t)st->W * count >= ((uint64_t)1 << 31))
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "SRS too large for 31-bit table indices");
    int rc;
    if ((rc = dev_alloc(c, &st->table, (size_t)st->W * count * sizeof(Affine<Q>)))) return rc;
    size_t m = (size_t)st->W * count;
    if ((rc = dev_alloc(c, (void**)&st->vals2, m * 4))) return rc;
    if ((rc = dev_alloc(c, &st->pairs, m * 8))) return rc;
    st->nb1 = (st->B >> MSM_BIN_LB) + 1;
    if (st->nb1 > (uint32_t)MSM_MAX_NB1) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "msm: too many level-1 bins");
    st->l1_scalars = (uint32_t)(MSM_L1_CAP / st->W) & ~63u;
    if (st->l1_scalars > 1024) st->l1_scalars = 1024;
    const size_t max_blk = (count + st->l1_scalars - 1) / st->l1_scalars;
    if ((rc = dev_alloc(c, (void**)&st->bin_offs, ((size_t)st->nb1 * max_blk + 1) * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->bin_aux, ((size_t)st->nb1 * max_blk / MSM_SCAN_TILE + 4) * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->bin_start, ((size_t)st->nb1 + 1) * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->tile_start, ((size_t)st->nb1 + 1) * 4))) return rc;
    st->l2_items = (uint32_t)(m / MSM_L2_TILE + st->nb1);
    if ((rc = dev_alloc(c, (void**)&st->cnt2, (size_t)st->l2_items * 256 * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->pos2, (size_t)st->l2_items * 256 * 4))) return rc;
    {
        const int lds = (int)(2 * st->nb1 * 4 + MSM_L1_CAP * 8);
        ZKT_HIP(c, hipFuncSetAttribute((const void*)k_msm_bin_scatter<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    static_assert(MsmState::SLOTS == 11, "per-slot arrays are sized for 11 slots");
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        if ((rc = dev_alloc(c, (void**)&st->offsets[i], ((size_t)st->nb1 * 256 + 2) * 4))) return rc;
        if ((rc = dev_alloc(c, (void**)&st->heavy[i], ((size_t)st->B + 2) * 4))) return rc;
    }
    {
        int blocks_per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_msm_accumulate<C>, 256, 0) == hipSuccess &&
            blocks_per_cu > 0 && cus > 0)
            st->acc_threads = (size_t)blocks_per_cu * cus * 256;
    }
    // chunk = max(ceil(pairs / acc_threads), MSM_CHUNK_MIN) pairs per thread, so an MSM never cuts its pairs into more
    // than acc_threads chunks (nor more than pairs / MSM_CHUNK_MIN): that bounds the piece array of every slot
    size_t max_chunks = std::min((m + MSM_CHUNK_MIN - 1) / MSM_CHUNK_MIN, st->acc_threads + 1);
    for (int i = 0; i < MsmState::SLOTS; ++i)
        if ((rc = dev_alloc(c, &st->pieces[i], (max_chunks + st->B + 2) * sizeof(XyzzRaw<Q>)))) return rc;
    if ((rc = dev_alloc(c, (void**)&st->chunk_bucket, (max_chunks + 2) * 4))) return rc;
    size_t nseg = st->B / MSM_SEG;
    ZKT_HIP(c, hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking));
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        if ((rc = dev_alloc(c, &st->buckets[i], ((size_t)st->B + 1) * sizeof(Xyzz<Q>)))) return rc;
        if ((rc = dev_alloc(c, &st->segA[i], nseg * sizeof(Xyzz<Q>)))) return rc;
        if ((rc = dev_alloc(c, &st->segT[i], nseg * sizeof(Xyzz<Q>)))) return rc;
        ZKT_HIP(c, hipHostMalloc(&st->host_result[i], (size_t)(MSM_MAX_Y + 1) * MSM_R2_BLOCKS * sizeof(Xyzz<Q>), hipHostMallocMapped));
        ZKT_HIP(c, hipHostGetDevicePointer(&st->host_result_dev[i], st->host_result[i], 0));
        ZKT_HIP(c, hipEventCreateWithFlags(&st->ev_main[i], hipEventDisableTiming));
        ZKT_HIP(c, hipEventCreateWithFlags(&st->ev_done[i], hipEventDisableTiming));
    }
    c->msm = st;
    ++c->msm_epoch;
    ++c->srs_generation;
    return ZKT_OK;
}

template <class C>
static int srs_finish(zkt_ctx* c) {
    using Q = typename C::Fq;
    MsmState& st = *c->msm;
    unsigned blocks = (unsigned)((st.count + 127) / 128);
    hipLaunchKernelGGL(k_srs_windows<C>, dim3(blocks), dim3(128), 0, c->stream, (Affine<Q>*)st.table, st.count, st.win);
    ZKT_HIP(c, hipGetLastError());
    const size_t total = (size_t)st.W * st.count;
    hipLaunchKernelGGL(k_srs_to_fx<C>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       (Affine<Q>*)st.table, total);
    ZKT_HIP(c, hipGetLastError());
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

void msm_release(zkt_ctx* c) {
    if (!c->msm) return;
    MsmState& st = *c->msm;
    (void)hipStreamSynchronize(c->stream);
    if (st.side) (void)hipStreamSynchronize(st.side);
    void* ptrs[] = {st.table,     st.vals2, st.pairs, st.bin_offs, st.bin_aux, st.bin_start,
                    st.tile_start, st.cnt2,   st.pos2,  st.chunk_bucket};
    for (void* p : ptrs) dev_free(c, p);
    for (int i = 0; i < MsmState::SLOTS; ++i) {
        dev_free(c, st.heavy[i]); dev_free(c, st.offsets[i]); dev_free(c, st.pieces[i]);
        dev_free(c, st.buckets[i]); dev_free(c, st.segA[i]); dev_free(c, st.segT[i]);
    }
    c->msm.reset();
}

template <class C>
static int srs_load_t(zkt_ctx* c, const void* src, size_t count, bool src_on_device, size_t slice_off = 0, size_t total = 0) {
    using Q = typename C::Fq;
    if (count == 0) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "empty SRS");
    msm_release(c);
    int rc = msm_setup<C>(c, count);
    if (rc) return rc;
    c->msm->slice_off = slice_off;
    c->msm->total = total ? total : count;
    ZKT_HIP(c, hipMemcpyAsync(c->msm->table, src, count * sizeof(Affine<Q>),
                              src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
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

// enqueue the whole MSM; its partial sums land in st.host_result[slot] (pinned), see msm_host_finish
template <class C>
static int msm_enqueue(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, int slot = 0) {
    using Q = typename C::Fq;
    using R = typename C::Fr;
    MsmState& st = *c->msm;
    ++c->msm_epoch;   // slot buffers change hands: anything issued ahead of time that relied on them is stale
    // the slot's buffers may still be read by the previous MSM that used this slot (side stream)
    if (st.pending[slot]) ZKT_HIP(c, hipStreamWaitEvent(c->stream, st.ev_done[slot], 0));
    const uint32_t m = (uint32_t)((size_t)st.W * n);
    // one chunk per thread, and exactly as many threads as the chip keeps resident for this kernel: the
    // whole array is consumed in a single wave-front with no partially filled last round
    uint32_t chunk = (uint32_t)((m + st.acc_threads - 1) / st.acc_threads);
    if (chunk < (uint32_t)MSM_CHUNK_MIN) chunk = MSM_CHUNK_MIN;
    {
    ProfScope prof_all(c, "msm_main");
    {
        const uint32_t S = st.l1_scalars;
        const unsigned nblk = (unsigned)((n + S - 1) / S);
        const uint32_t total = st.nb1 * nblk, ntiles = (total + MSM_SCAN_TILE - 1) / MSM_SCAN_TILE;
        hipLaunchKernelGGL(k_msm_bin_count<C>, dim3(nblk), dim3(1024), (size_t)st.nb1 * 4, c->stream,
                           (const Fe<R>*)d_scalars, n, mont, st.win, S, st.nb1, st.bin_offs);
        hipLaunchKernelGGL(k_msm_scan_tiles, dim3(ntiles), dim3(1024), 0, c->stream, st.bin_offs, total, st.bin_aux);
        hipLaunchKernelGGL(k_msm_scan_aux, dim3(1), dim3(1024), 0, c->stream, st.bin_offs, st.bin_aux, ntiles, nblk, st.nb1,
                           st.bin_start, st.tile_start, st.heavy[slot]);
        ZKT_HIP(c, hipGetLastError());
        hipLaunchKernelGGL(k_msm_bin_scatter<C>, dim3(nblk), dim3(1024), (size_t)st.nb1 * 8 + (size_t)MSM_L1_CAP * 8,
                           c->stream, (const Fe<R>*)d_scalars, n, mont, st.win, S, st.count, base_off, st.nb1,
                           st.bin_offs, st.bin_aux, (uint2*)st.pairs);
        ZKT_HIP(c, hipGetLastError());
        const uint32_t items = (uint32_t)(m / MSM_L2_TILE + st.nb1);
        hipLaunchKernelGGL(k_msm_l2_count, dim3(items), dim3(256), 0, c->stream, (const uint2*)st.pairs, st.nb1,
                           st.bin_start, st.tile_start, st.cnt2);
        hipLaunchKernelGGL(k_msm_l2_scan, dim3(st.nb1), dim3(256), 0, c->stream, st.cnt2, st.pos2, st.bin_start,
                           st.tile_start, st.offsets[slot], st.B, chunk, st.chunk_bucket);
        hipLaunchKernelGGL(k_msm_l2_scatter, dim3(items), dim3(256), 0, c->stream, (const uint2*)st.pairs, st.nb1,
                           st.bin_start, st.tile_start, st.cnt2, st.pos2, st.vals2);
        ZKT_HIP(c, hipGetLastError());
    }
    {
        ProfScope prof_acc(c, "msm_accumulate");
        uint32_t max_chunks = (m + chunk - 1) / chunk;
        hipLaunchKernelGGL(k_msm_accumulate<C>, dim3((max_chunks + 255) / 256), dim3(256), 0, c->stream, st.vals2,
                           st.B, chunk, st.offsets[slot], st.chunk_bucket, (const Affine<Q>*)st.table, (XyzzRaw<Q>*)st.pieces[slot]);
        ZKT_HIP(c, hipGetLastError());
    }
    }
    // ---- tail on the side stream: the bucket fold (latency bound: one wave per SIMD, three dependent additions) and the
    // bucket reduction overlap whatever the main stream does next; everything they read is the slot's own ----
    ZKT_HIP(c, hipEventRecord(st.ev_main[slot], c->stream));
    ZKT_HIP(c, hipStreamWaitEvent(st.side, st.ev_main[slot], 0));
    {
    ProfScope prof_fold(c, "msm_fold", st.side);
    hipLaunchKernelGGL(k_msm_bucket_sum<C>, dim3((st.B + 1 + 255) / 256), dim3(256), 0, st.side, st.offsets[slot], st.B,
                       chunk, (const XyzzRaw<Q>*)st.pieces[slot], (Xyzz<Q>*)st.buckets[slot], st.heavy[slot]);
    ZKT_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_msm_heavy<C>, dim3(MSM_HEAVY_BLOCKS), dim3(256), 0, st.side, st.offsets[slot], chunk,
                       (const XyzzRaw<Q>*)st.pieces[slot], (Xyzz<Q>*)st.buckets[slot], st.heavy[slot]);
    ZKT_HIP(c, hipGetLastError());
    }
    int ny = 0;
    {
    ProfScope prof_tail(c, "msm_tail", st.side);
    const uint32_t nseg = st.B / MSM_SEG;
    hipLaunchKernelGGL(k_msm_segments<C>, dim3((nseg + 255) / 256), dim3(256), 0, st.side,
                       (const Xyzz<Q>*)st.buckets[slot], nseg, (Xyzz<Q>*)st.segA[slot], (Xyzz<Q>*)st.segT[slot]);
    ZKT_HIP(c, hipGetLastError());
    ny = msm_rows(st.c);
    hipLaunchKernelGGL(k_msm_masked_sums<C>, dim3(MSM_R2_BLOCKS, ny + 1), dim3(256), 0, st.side,
                       (const Xyzz<Q>*)st.segA[slot], (const Xyzz<Q>*)st.segT[slot], nseg, ny,
                       (const Xyzz<Q>*)st.buckets[slot] + st.B, (Xyzz<Q>*)st.host_result_dev[slot]);
    ZKT_HIP(c, hipGetLastError());
    }
    // the (ny + 1) x R2_BLOCKS partial sums are written straight into pinned host memory (129 posted writes of 128 B;
    // a copy engine took ~90 us for them): the host finishes the reduction (msm_host_finish) once ev_done has fired
    ZKT_HIP(c, hipEventRecord(st.ev_done[slot], st.side));
    st.pending[slot] = true;
    return ZKT_OK;
}

// S = sum_y 2^(e_y) V_y : V_0 = sum of segA (e = 0), V_(k+1) = sum of segT over segments with bit k (e = k + LOG_SEG),
// the top bucket (e = c - 1).  ~150 curve operations on 64-bit limbs.
template <class Q>
static Xyzz<Q> msm_host_finish(const MsmState& st, int slot) {
    const int ny = msm_rows(st.c);
    int exps[MSM_MAX_Y + 1];
    exps[0] = 0;
    for (int y = 1; y < ny; ++y) exps[y] = (y - 1) + MSM_LOG_SEG;
    exps[ny] = st.c - 1;
    return hostec::weighted_row_sum<Q>((const Xyzz<Q>*)st.host_result[slot], ny + 1, MSM_R2_BLOCKS, exps);
}

// waits for the MSM in `slot` and normalises its result on the host (one inversion; the affine
// coordinates are needed there for the Fiat-Shamir transcript anyway)
template <class C>
static int msm_collect(zkt_ctx* c, int slot, Affine<typename C::Fq>* out) {
    using Q = typename C::Fq;
    MsmState& st = *c->msm;
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
int msm_begin(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, int slot) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (slot < 0 || slot >= MsmState::SLOTS) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "bad MSM slot");
    if (n == 0 || base_off > c->msm->count || n > c->msm->count - base_off)
        return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
    if (c->curve == ZKT_CURVE_BN254) return msm_enqueue<Bn254Curve>(c, d_scalars, n, base_off, mont, slot);
    return msm_enqueue<Bls381Curve>(c, d_scalars, n, base_off, mont, slot);
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
template <class Q>
static void table_to_ark(Affine<Q>* pts, size_t count) {
    for (size_t i = 0; i < count; ++i) {
        if (aff_is_inf<Q>(pts[i])) continue;
        pts[i].x = fx_to_ark<Q>(fx_unpack<Q>(pts[i].x));
        pts[i].y = fx_to_ark<Q>(fx_unpack<Q>(pts[i].y));
    }
}

int srs_download(zkt_ctx* c, size_t offset, size_t count, uint64_t* out) {
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded");
    if (offset > c->msm->count || count > c->msm->count - offset) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "range");
    size_t psz = (c->curve == ZKT_CURVE_BN254) ? sizeof(Affine<Bn254Fq>) : sizeof(Affine<Bls381Fq>);
    ZKT_HIP(c, hipMemcpyAsync(out, (const char*)c->msm->table + offset * psz, count * psz, hipMemcpyDeviceToHost,
                              c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    // the device table is kept in R' = 2^(29 L) Montgomery form; hand back arkworks' R form (test / bench aid)
    if (c->curve == ZKT_CURVE_BN254) table_to_ark<Bn254Fq>((Affine<Bn254Fq>*)out, count);
    else table_to_ark<Bls381Fq>((Affine<Bls381Fq>*)out, count);
    return ZKT_OK;
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
