// State of the fixed-base MSM (msm.hip) shared with the Lagrange-basis table builder (lagrange.hip).
#pragma once
#include "ctx.hpp"

namespace zkt {

// Grouped pairs per accumulation thread ("chunk").  It is chosen ON THE DEVICE from the number of pairs an MSM really has
// (zero digits are dropped, so a sparse scalar vector -- the differences of a piecewise-constant evaluation vector,
// lagrange.hip -- has few): chunk = max(ceil(pairs / acc_threads), lo), lo = isqrt(2 pairs / buckets) clamped to
// [1, MSM_CHUNK_MIN]: the length of a thread's serial chain and the pieces per bucket the fold has to add are balanced.
constexpr int MSM_CHUNK_MIN = 16;
constexpr int MSM_R2_BLOCKS = 1;     // partial sums per row handed to the host
constexpr int MSM_MAX_Y = 24;

// Window layout: W windows of width c or c-1 covering exactly lambda+1 bits, so that no window
// (in particular not the top one) is left with only a few significant bits: a 2-bit top window
// would pour n entries into 4 buckets.
struct MsmWindows {
    int W;
    uint8_t width[40];
    uint16_t start[40];
};

constexpr int MSM_BATCH = 3;         // MSMs whose grouping and accumulation go out as ONE launch per kernel (blockIdx.y = MSM)
// Per-MSM arguments of a batched launch.  The main-stream work buffers (pairs, grouped indices, count tables) exist
// MSM_BATCH times, `s_*` apart; what the side stream reads later (offsets, pieces, params, heavy list) belongs to a slot.
struct MsmBatch {
    const void* scalars[MSM_BATCH];
    uint64_t n[MSM_BATCH];
    uint64_t base_off[MSM_BATCH];
    const void* table[MSM_BATCH];    // the base table each MSM runs against (the key's powers or the Lagrange-prefix table)
    uint64_t tcount[MSM_BATCH];      // ... and its bases per window
    uint32_t* heavy[MSM_BATCH];
    uint32_t* params[MSM_BATCH];
    uint32_t* offsets[MSM_BATCH];
    void* pieces[MSM_BATCH];
    uint64_t s_bin_offs, s_bin_aux, s_bin, s_tile_desc, s_cnt, s_pairs_bytes, s_vals, s_chunk;   // strides, in elements (pairs: bytes)
};

constexpr int MSM_TAIL_BATCH = 6;    // bucket folds / reductions that go out as ONE launch per kernel (blockIdx.y = slot)
struct MsmTailBatch {                // per-slot arguments of a batched tail
    const uint32_t* offsets[MSM_TAIL_BATCH];
    const uint32_t* params[MSM_TAIL_BATCH];
    const void* pieces[MSM_TAIL_BATCH];
    void* buckets[MSM_TAIL_BATCH];
    uint32_t* heavy[MSM_TAIL_BATCH];
    void* rowcol[MSM_TAIL_BATCH];
    void* partials[MSM_TAIL_BATCH];
};

constexpr int MSM_HEAVY = 32;       // buckets with more pieces than this are folded by a whole block
constexpr int MSM_HEAVY_BLOCKS = 64;     // grid-stride over the (normally empty) list of crowded buckets

struct MsmState {
    size_t count = 0;      // bases loaded
    // index-range sharding: this GPU holds powers [slice_off, slice_off + count) of a key of `total`
    size_t slice_off = 0, total = 0;
    int c = 0, W = 0;      // max window bits, windows
    MsmWindows win{};
    uint32_t* heavy[11] = {};   // per slot: [0] = count, [1..] = heavy bucket ids
    uint32_t B = 0;        // buckets = 2^(c-1), ids 1..B
    void* table = nullptr; // Affine[W][count]
    // Second base table (lagrange.hip): prefix sums of the Lagrange-basis key of the domain of size 2^lag_log_n followed by
    // the blinder points, same window layout, count2 <= count bases.  Commitments of polynomials given by their
    // evaluations go through it (msm_begin table = 1).  Null until a circuit of that size asks for it.
    void* table2 = nullptr;
    size_t count2 = 0;
    int lag_log_n = -1;
    bool lag_failed = false;       // the key is too short (or sharded): evaluations are committed through their coefficients
    // zkt_ctx_fork: the base tables belong to the context this one was forked from (read-only here, never freed here)
    bool table_borrowed = false, table2_borrowed = false;
    uint32_t* params[11] = {};     // per slot, device: [0] chunk, [1] pairs (written by k_msm_scan_aux)
    // work buffers (sized for n = count)
    uint32_t* vals2 = nullptr;                     // table indices grouped by bucket
    void* pairs = nullptr;                         // uint2[m]: after the level-1 split
    uint32_t* bin_offs = nullptr;                  // [nb1][blocks] level-1 counts, scanned per 4096-tile
    uint32_t* bin_aux = nullptr;                   // tile totals, scanned; last = number of pairs
    uint32_t *bin_start = nullptr, *tile_start = nullptr;   // nb1 + 1 each: level-2 work list
    void* tile_desc = nullptr;                     // uint2[l2_items]: pair range of every level-2 tile
    uint32_t *cnt2 = nullptr, *pos2 = nullptr;     // [level-2 tiles][256]
    uint32_t* chunk_bucket = nullptr;              // bucket of the first pair of every accumulation chunk
    uint32_t nb1 = 0;                              // level-1 bins
    uint32_t lb = 8;                               // level-2 key bits: bucket = (bin << lb) | low
    int lcols = 8;                                 // log2 columns of the level-2 tables (8, or 10 for more than 2^17 buckets)
    bool packed = false;                           // (low key, table index, sign) fit ONE 32-bit word: 4-byte pairs
    int dig = 0;                                   // compile-time window layout of the level-1 kernels (0: generic)
    uint32_t l1_scalars = 0;                       // scalars per level-1 workgroup
    uint32_t l2_items = 0;                         // upper bound of level-2 tiles
    MsmBatch strides{};                            // the s_* members: distance between the work buffers of a batch's MSMs
    // per slot, because the bucket fold that reads them runs on the side stream while the next MSM is already grouping
    uint32_t* offsets[11] = {};   // B + 2
    void* pieces[11] = {};        // XyzzRaw[max_chunks + B + 2]
    // The latency-bound tail of an MSM (bucket reduction) runs on a side stream so that it overlaps the
    // next MSM's accumulation; each in-flight MSM owns one slot of tail buffers.
    static constexpr int SLOTS = 11;
    void* buckets[SLOTS] = {};      // Xyzz[B + 1]
    void* rowcol[SLOTS] = {};       // Xyzz[NI + NJ]: row / column sums of the bucket matrix
    void* host_result[SLOTS] = {};  // pinned: the (rows + 1) x R2_BLOCKS partial sums the host finishes
    void* host_result_dev[SLOTS] = {};  // the same memory as the kernels address it
    size_t acc_lds = 0;            // dynamic LDS of k_msm_accumulate (0; ZKT_MSM_ACC_LDS caps its residency in experiments)
    size_t acc_threads = 196608;   // chunks an MSM is cut into: resident threads of k_msm_accumulate (occupancy query) x 2
    hipStream_t side = nullptr;
    hipEvent_t ev_main[SLOTS] = {}, ev_done[SLOTS] = {};
    bool pending[SLOTS] = {};
    // Small keys (count <= MSM_DEFER_MAX): a proof is then a chain of latencies, and the bucket reduction of every
    // commitment of a round is the same ~25 dependent curve operations whether one launch sequence covers one slot or six:
    // the tails are deferred and issued once per round, batched (msm_flush_tails).  Large keys keep a tail per enqueue: it
    // overlaps the next commitment's accumulation instead of the transforms behind the round.
    bool defer_tails = false;
    int tail_wait[SLOTS] = {};     // slots whose accumulation is enqueued and whose tail is not, in order
    int n_tail_wait = 0;
    ~MsmState() {
        for (int i = 0; i < SLOTS; ++i) {
            if (host_result[i]) (void)hipHostFree(host_result[i]);
            if (ev_main[i]) (void)hipEventDestroy(ev_main[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
        }
        if (side) (void)hipStreamDestroy(side);
    }
};


// msm.hip: window multiples + R' conversion of an affine base table whose first `count` entries are filled (arkworks R form)
int msm_table_finish(zkt_ctx* c, void* table, size_t count);
// msm.hip: `child` gets an MSM state of its own (work buffers, slots, side stream) over `parent`'s base tables
int msm_fork(zkt_ctx* child, const zkt_ctx* parent);
// issues the deferred tails (no-op when none are waiting); the prover calls it behind the last commitment of a round
int msm_flush_tails(zkt_ctx* c);
bool msm_defers_tails(const zkt_ctx* c);
bool msm_batches_grouping(const zkt_ctx* c);   // a round's commitments are grouped as one batch of launches (small keys, and 2^18)
constexpr size_t MSM_DEFER_MAX = ((size_t)1 << 16) + 64;      // small key: latency regime (tails deferred and batched)
constexpr size_t MSM_TAIL_INL_MAX = ((size_t)1 << 18) + 64;   // up to here the bucket reduction's additions inline their products

}  // namespace zkt
