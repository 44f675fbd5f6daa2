// Device-resident PLONK+Plookup prover: host orchestration of the five rounds of
// plonk-core/src/proof_system/prove.rs:59-470.  Witness vectors are uploaded once; every
// polynomial stays in HBM between rounds; only commitments (affine points), the 12 evaluations and
// two scalars (the grand-product denominators) cross PCIe, for the host-side Fiat-Shamir transcript.
#include "ctx.hpp"
#include "hostinv.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "ec.hpp"
#include "keyfile.hpp"
#include "poly.hpp"
#include "transcript.hpp"

#include <algorithm>
#include <cstring>
#include <memory>
#include <optional>
#include <vector>

namespace zkt {

// msm.hip (tbl = 1: the Lagrange-prefix table of lagrange.hip)
int msm_g1_dev(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, uint64_t* out_xy, int* out_inf);
int msm_begin(zkt_ctx* c, const void* d_scalars, size_t n, size_t base_off, int mont, int slot, int tbl = 0);
int msm_begin_batch(zkt_ctx* c, int k, const void* const* d_scalars, const size_t* ns, int mont, const int* slots, const int* tbls);
int msm_flush_tails(zkt_ctx* c);           // issues the deferred bucket reductions of the commitments begun so far (small keys)
bool msm_defers_tails(const zkt_ctx* c);
bool msm_batches_grouping(const zkt_ctx* c);
// lagrange.hip
int lagrange_ensure(zkt_ctx* c, int log_n);
bool lagrange_ready(const zkt_ctx* c, int log_n);
size_t lagrange_bases(const zkt_ctx* c);
int msm_end(zkt_ctx* c, int slot, uint64_t* out_xy);
int msm_end_sharded(zkt_ctx* c, const int* slots, const bool* have, int k, uint64_t* out_xy);
void msm_slice(zkt_ctx* c, size_t* off, size_t* count, size_t* total);

enum { PK_QM = 0, PK_QL, PK_QR, PK_QO, PK_QC, PK_S1, PK_S2, PK_S3, PK_QLOOKUP, PK_QTABLE, PK_COUNT };
// coset vectors kept on the device (keys/mod.rs:153-174; x and l_1 are stored, zh is 4 scalars)
enum { CS_QM = 0, CS_QL, CS_QR, CS_QO, CS_QC, CS_QLOOKUP, CS_QTABLE, CS_S1, CS_S2, CS_S3, CS_X, CS_L1, CS_COUNT };
enum { W_A = 0, W_B, W_C, W_PI, W_Z1, W_Z2, W_T, W_H1, W_H2, W_COUNT };  // witness cosets of quotient_poly.rs:52-96

struct CircuitState {
    int log_n = 0;
    size_t n = 0;
    // A proof sharded over G GPUs (zkt_ctx_set_comm): this GPU owns the 4n-coset points of global index cls modulo G,
    // m = 4n / G of them; coset[] and wcos[] hold that class only.  G = 1: the whole coset, as the reference has it.
    int G = 1, cls = 0, log_m = 0;
    size_t m = 0;
    void* wnext[4] = {};       // G = 8: z1, z2, t, h1 on the class (cls + 4) mod 8, where "omega-next" lives
    void* fold = nullptr;      // m elements: a polynomial folded modulo X^m - shift^m before its class transform
    void* qgather = nullptr;   // 4n elements: every rank's quotient evaluations, class-major (the all-gather's target)
    bool have[16] = {};        // commitment slot -> this rank's SRS slice takes part (an MSM was started)
    void* pk[PK_COUNT] = {};
    size_t pk_len[PK_COUNT] = {};
    void* coset[CS_COUNT] = {};
    void* sigma_ev[3] = {};
    void* q_lookup_ev = nullptr;
    void* roots = nullptr;
    uint32_t zh_inv[4][8] = {};
    // per-proof work buffers
    void* ev[8] = {};       // a, b, c, t, f, h1, h2, pi   (n each)
    void* sc[4] = {};       // scan scratch: num, den, PN, SD (n each)
    void* scan_tmp = nullptr;
    void* poly[13] = {};    // a b c t h1 h2 z1 z2 pi q_lo q_mid q_hi work   (n + 8 each)
    void* wcos[W_COUNT] = {};
    void* qev = nullptr;    // 4n
    // scalars of a commitment taken in the Lagrange basis (read by the MSM's level-1 kernels only): n + 16 each; one per
    // commitment that may wait in the queue of a round (t, h1, h2; z2), the last also stages zkt_commit_evals_dev's blinders
    void* lag_scalars = nullptr;
    void* lag_scalars_q[3] = {};
    void* small = nullptr;  // blinders (19), eval partials, eval results
    uint32_t* status = nullptr;  // [0] error bits, [1] len scratch ... [4..7] quotient lens, [8..] poly lens
    uint32_t* lk_u32 = nullptr;  // lookup: perm, base counts, starts of the even half, of the odd half, hit counts
    uint32_t lk_nkeys = 0;       // keys of the resident lookup tables (0 = none)
    std::vector<uint32_t> lk_host_keys, lk_host_sorted, lk_host_counts, lk_host_order;   // host staging (8 words per key)
    // zkt_prove_set_next: rounds 1 and 2 of the NEXT proof depend on no challenge; they are issued behind the last
    // commitments of the current proof, so the GPU never drains between two proofs
    // Round 1 of the next proof goes behind the quotient commitments of round 4, round 2 behind the opening commitments
    // of round 5.  Round 1 overwrites what round 5 of the current proof still reads (the wire polynomials) and both
    // use the status words, so those exist twice; `*_alt` is the set the early work writes.
    bool has_next = false, prefetch_same_table = false;
    int prefetch_stage = 0;        // 0 nothing, 1 round 1 issued, 2 rounds 1 and 2 issued (only then it is usable)
    uint64_t prefetch_epoch = 0;   // zkt_ctx::msm_epoch right after the early work was issued
    zkt_prove_inputs next_in{}, prefetch_in{};
    void* poly_alt[3] = {};
    uint32_t* status_alt = nullptr;
    void* lk_keys = nullptr;     // insertion-order keys then sorted keys
    size_t lk_cap = 0;
    void* pinned = nullptr;
    hipStream_t comm_stream = nullptr;   // sharded proof, asynchronous communicator: the quotient exchange's pieces travel here
    hipEvent_t ev_piece[ZKT_QUOTIENT_CHUNKS] = {}, ev_gathered = nullptr;
    // Round 5: the second opening's polynomial work (a linear combination and a division: memory-bound) runs on this stream
    // beside the first opening's commitment (integer-ALU bound), in buffers of its own
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux_go = nullptr, ev_aux_done = nullptr;
    void* aux_scan_tmp = nullptr;
    void* aux_pw = nullptr;
    hipStream_t copy_stream = nullptr;   // host witness uploads travel beside the main stream's work (cold path)
    hipEvent_t ev_copy[4] = {};          // a, b, c uploaded; [3]: main stream reached the upload point
    void* pinned_pi = nullptr;      // host staging of pi_tab
    uint32_t* pi_tab = nullptr;     // device: QUOTIENT_PI_DIRECT_MAX entries of {rotation, 9 limbs}
    void* eval_pw = nullptr;        // EVAL_MAX x 257 powers of the evaluation points (round 5)
    // The table polynomial depends on the lookup table only: while consecutive proofs pass the same table its
    // evaluations, coefficients, 4n-coset and commitment are reused (one MSM and two transforms less).
    std::vector<uint64_t> cached_table;
    bool t_cached = false, t_coset_valid = false;
    uint64_t t_srs_generation = 0;   // zkt_ctx::srs_generation the cached commitment was made under
    uint64_t t_commit_xy[12] = {};
    // zkt_ctx_fork: the keys (pk, coset, sigma_ev, q_lookup_ev, roots) belong to the context this one was forked from
    bool keys_borrowed = false;
};

// ---- host field helpers ------------------------------------------------------------------------------
template <class P>
struct HostF {
    using F = Fe<P>;
    static F from_words(const uint64_t* w) {
        F r;
        memcpy(r.v, w, 32);
        return r;
    }
    static void to_le_bytes(const F& mont, uint8_t out[32]) {
        F c = fe_from_mont<P>(mont);
        memcpy(out, c.v, 32);
    }
    static F from_le_bytes(const uint8_t in[32]) {
        F c;
        memcpy(c.v, in, 32);
        return fe_to_mont<P>(c);
    }
    static F pow_u64(F a, uint64_t e) { return fe_pow_u64<P>(a, e); }
};

template <class Q>
static void fq_to_le(const Fe<Q>& mont, uint8_t* out) {
    Fe<Q> c = fe_from_mont<Q>(mont);
    memcpy(out, c.v, Q::N * 4);
}

// ark-serialize 0.3 compressed short-Weierstrass point (proof wire format, proof.rs:98-155):
// x little-endian, bit 7 of the last byte = y > -y, bit 6 = infinity.
template <class Q>
static void serialize_point(const Affine<Q>& p, std::vector<uint8_t>& out) {
    const size_t nb = Q::N * 4;
    std::vector<uint8_t> b(nb, 0);
    if (aff_is_inf<Q>(p)) {
        b[nb - 1] |= 0x40;
    } else {
        fq_to_le<Q>(p.x, b.data());
        Fe<Q> y = fe_from_mont<Q>(p.y);
        Fe<Q> ny = fe_from_mont<Q>(fe_neg<Q>(p.y));
        bool greater = false;
        for (int i = Q::N - 1; i >= 0; --i) {
            if (y.v[i] != ny.v[i]) {
                greater = y.v[i] > ny.v[i];
                break;
            }
        }
        if (greater) b[nb - 1] |= 0x80;
    }
    out.insert(out.end(), b.begin(), b.end());
}

template <class C>
struct Prover {
    using R = typename C::Fr;
    using Q = typename C::Fq;
    using F = Fe<R>;
    using H = HostF<R>;

    zkt_ctx* c;
    CircuitState& S;
    HostTranscript& tr;
    Prover(zkt_ctx* ctx, CircuitState& st, HostTranscript& t) : c(ctx), S(st), tr(t) {
        trace_on = exp_env("ZKT_HOST_TRACE") != nullptr;
    }
    bool copy_fenced = false;   // the copy stream already waits for the main stream's earlier work (run(), cold path)
    // ZKT_HOST_TRACE=1: wall-clock marks of the host control path (where the GPU may be waiting for the host)
    bool trace_on = false;
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    void mark(const char* what) {
        if (trace_on) marks.emplace_back(what, std::chrono::steady_clock::now());
    }
    void dump_marks() {
        if (!trace_on || marks.empty()) return;
        for (size_t i = 1; i < marks.size(); ++i)
            fprintf(stderr, "[zkt host] %-28s %8.1f us\n", marks[i].first,
                    std::chrono::duration<double, std::micro>(marks[i].second - marks[i - 1].second).count());
        marks.clear();
    }

    void tr_commit(const char* label, const Affine<Q>& p) {
        uint8_t x[64], y[64];
        bool inf = aff_is_inf<Q>(p);
        if (!inf) {
            fq_to_le<Q>(p.x, x);
            fq_to_le<Q>(p.y, y);
        }
        tr.append_commitment(label, x, y, Q::N * 4, inf);
    }
    void tr_scalar(const char* label, const F& v) {
        uint8_t b[32];
        H::to_le_bytes(v, b);
        tr.append_scalars(label, b, 1, 32, true);
    }
    F tr_challenge(const char* label) {
        uint8_t b[32];
        tr.challenge_scalar(label, R::BITS, b);
        return H::from_le_bytes(b);
    }
    static void put(uint32_t dst[8], const F& v) { memcpy(dst, v.v, 32); }

    // Idle time of the main stream across a host round trip (zkt_profile_get "host_wait"): the first event is recorded
    // behind everything enqueued when the host starts to wait (it fires when the stream drains), the second right
    // before the next launch.  bench.py's gpu_active is the wall clock minus these.
    std::optional<ProfScope> idle;
    void wait_begin() {
        if (c->prof_on && !idle) idle.emplace(c, "host_wait");
    }
    void wait_end() { idle.reset(); }

    // commitments of one round are enqueued back to back (the bucket-reduction tail of one overlaps the
    // accumulation of the next) and collected together
    // With the committer key sharded by index range, a commitment is this rank's partial sum over
    // coefficients [slice_off, slice_off + slice_count) -- nothing at all when the polynomial ends below the slice.
    int commit_begin(const void* d_poly, size_t len, int slot) {
        if (!c->sharded()) return msm_begin(c, d_poly, len, 0, 1, slot);
        size_t off, cnt, total;
        msm_slice(c, &off, &cnt, &total);
        if (len > total)
            return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
        S.have[slot] = len > off;
        if (!S.have[slot]) return ZKT_OK;
        const size_t l = std::min(len, off + cnt) - off;
        return msm_begin(c, (const char*)d_poly + off * 32, l, 0, 1, slot);
    }
    // Commitments wait in a queue until the round has begun them all (commit_flush), then go out in batches of up to
    // MSM_BATCH launches-as-one (msm.hip msm_enqueue_batch: blockIdx.y = MSM, either base table per entry).  Large and sharded
    // keys issue every commitment at once instead (msm_batches_grouping says which sizes gain).
    struct Pending {
        const void* scalars;
        size_t len;
        int slot, tbl;
    };
    Pending queue[6];
    int n_queue = 0, n_lag_queued = 0;
    bool queueing() const { return !c->sharded() && !c->batch_off && msm_batches_grouping(c); }
    int commit_flush() {
        int rc = ZKT_OK;
        for (int at = 0; at < n_queue && !rc; at += 3) {
            const int k = std::min(3, n_queue - at);
            const void* sc[3];
            size_t lens[3];
            int slots[3], tbls[3];
            for (int j = 0; j < k; ++j) {
                sc[j] = queue[at + j].scalars; lens[j] = queue[at + j].len; slots[j] = queue[at + j].slot; tbls[j] = queue[at + j].tbl;
            }
            rc = msm_begin_batch(c, k, sc, lens, 1, slots, tbls);
        }
        n_queue = 0;
        n_lag_queued = 0;
        if (rc) return rc;
        return msm_flush_tails(c);
    }
    int commit_push(const void* scalars, size_t len, int slot, int tbl) {
        if (n_queue == 6)
            if (int rc = commit_flush()) return rc;
        queue[n_queue++] = Pending{scalars, len, slot, tbl};
        return ZKT_OK;
    }
    int commit_begin_many(void* const* d_polys, const size_t* lens, const int* slots, int k) {
        for (int j = 0; j < k; ++j) {
            const int rc = queueing() ? commit_push(d_polys[j], lens[j], slots[j], 0) : commit_begin(d_polys[j], lens[j], slots[j]);
            if (rc) return rc;
        }
        return ZKT_OK;
    }
    // Commitment of the polynomial with evaluations `ev` plus k blinders (prove.rs:166-180,249-251), whose blinded
    // coefficients are in d_poly.  With the Lagrange-basis table (lagrange.hip) the scalars are the differences of
    // neighbouring evaluations: for t, h1, h2 and z2 all but a few thousand of them are zero and the MSM costs next to
    // nothing; without it (sharded key, key shorter than n + 1) the coefficients are committed as the reference does.
    int commit_evals_begin(const void* ev, const void* d_poly, int blinder_off, int k, int len_slot, int slot) {
        const bool q = queueing();
        if (!lagrange_ready(c, S.log_n) || c->lagrange_off)
            return q ? commit_push(d_poly, S.n + (size_t)k, slot, 0) : commit_begin(d_poly, S.n + (size_t)k, slot);
        if (q && n_lag_queued == 3)
            if (int rc0 = commit_flush()) return rc0;
        void* d = q ? S.lag_scalars_q[n_lag_queued] : S.lag_scalars;   // a queued commitment keeps its scalars until the flush
        int rc = lagrange_scalars(c, ev, S.n, S.status + 8 + len_slot, (const char*)S.small + (size_t)blinder_off * 32, k, S.roots, d);
        if (rc) return rc;
        if (!q) return msm_begin(c, d, S.n + (size_t)k, 0, 1, slot, 1);
        ++n_lag_queued;
        return commit_push(d, S.n + (size_t)k, slot, 1);
    }
    // The commitments of one prover round, collected together; skip[j]: nothing was started for entry j (out[j] is left
    // alone).  Sharded: ONE all-gather of the round's partial sums (msm.hip msm_collect_sharded).
    int commit_collect(const int* slots, const bool* skip, int k, Affine<Q>* out) {
        wait_begin();
        if (!c->sharded()) {
            for (int j = 0; j < k; ++j) {
                if (skip && skip[j]) continue;
                uint64_t xy[12];
                int rc = msm_end(c, slots[j], xy);
                if (rc) return rc;
                memcpy(out[j].x.v, xy, Q::N * 4);
                memcpy(out[j].y.v, xy + Q::N / 2, Q::N * 4);
            }
            return ZKT_OK;
        }
        bool have[16];
        uint64_t xy[16 * 12];
        for (int j = 0; j < k; ++j) have[j] = !(skip && skip[j]) && S.have[slots[j]];
        int rc = msm_end_sharded(c, slots, have, k, xy);   // the message has k entries on every rank, whatever `have` says
        if (rc) return rc;
        for (int j = 0; j < k; ++j) {
            if (skip && skip[j]) continue;
            memcpy(out[j].x.v, xy + 12 * j, Q::N * 4);
            memcpy(out[j].y.v, xy + 12 * j + Q::N / 2, Q::N * 4);
        }
        return ZKT_OK;
    }

    // iNTT of n evaluations into a zero-tailed coefficient buffer, trim, blind (prove.rs:120-127 etc.); the polynomials
    // of one round go through the transform as ONE batch (one launch per pass)
    struct PolyJob {
        const void* ev;
        void* poly;
        int blinder_off, k, len_slot;
    };
    int evals_to_blinded_polys(const PolyJob* jobs, int nb) {
        const size_t n = S.n;
        int rc;
        const void* ins[NTT_MAX_BATCH];
        void* outs[NTT_MAX_BATCH];
        size_t lens[NTT_MAX_BATCH];
        for (int y = 0; y < nb; ++y) {
            ins[y] = jobs[y].ev;
            outs[y] = jobs[y].poly;
            lens[y] = n;
        }
        if ((rc = ntt_run_batch(c, S.log_n, 1, 0, nb, ins, lens, outs))) return rc;
        for (int y = 0; y < nb; ++y) {
            void* poly = jobs[y].poly;
            if (jobs[y].k > 0) {
                if ((rc = poly_trim_len(c, poly, n, S.status + 8 + jobs[y].len_slot, (char*)poly + n * 32, 8))) return rc;
                if ((rc = poly_add_blinders(c, poly, S.status + 8 + jobs[y].len_slot,
                                            (const char*)S.small + (size_t)jobs[y].blinder_off * 32, jobs[y].k, n + 8)))
                    return rc;
            } else {
                ZKT_HIP(c, hipMemsetAsync((char*)poly + n * 32, 0, 8 * 32, c->stream));
            }
        }
        return ZKT_OK;
    }
    int evals_to_blinded_poly(const void* ev, void* poly, int blinder_off, int k, int len_slot) {
        const PolyJob j{ev, poly, blinder_off, k, len_slot};
        return evals_to_blinded_polys(&j, 1);
    }

    int check_status() {
        uint32_t st = 0;
        ZKT_HIP(c, hipMemcpyAsync(&st, S.status, 4, hipMemcpyDeviceToHost, c->stream));
        wait_begin();
        ZKT_HIP(c, hipStreamSynchronize(c->stream));
        if (st & 16u) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "wire index outside the variable map");
        if (st & 4u) return set_err(c, ZKT_ERR_NOT_IN_TABLE, "ElementNotIndexedInTable: a looked-up value is not in the table");
        if (st & 8u) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "combine_split: h1/h2 length differs from n");
        if (st & 2u) return set_err(c, ZKT_ERR_QUOTIENT_TOO_SHORT, "quotient degree exceeds 3n+5: the circuit is not satisfied");
        if (st & 1u) return set_err(c, ZKT_ERR_QUOTIENT_TOO_SHORT, "quotient polynomial too short to split (prove.rs:287-300)");
        return ZKT_OK;
    }

    // lookup/multiset.rs:103-146 on the device; table = distinct values in insertion order
    // `fresh` = the lookup table differs from the previous proof's: its keys (insertion order and sorted), the
    // sort permutation and the base multiplicities are rebuilt and uploaded; otherwise they are still resident.
    int combine_split(const uint64_t* table, size_t table_len, bool fresh) {
        const size_t n = S.n;
        F* d_keys = (F*)S.lk_keys;
        F* d_sorted = d_keys + S.lk_cap;
        uint32_t* d_perm = S.lk_u32;
        uint32_t* d_base = S.lk_u32 + S.lk_cap;
        uint32_t* d_even = S.lk_u32 + 2 * S.lk_cap;
        uint32_t* d_odd = S.lk_u32 + 3 * S.lk_cap;
        uint32_t* d_hits = S.lk_u32 + 4 * S.lk_cap;   // zero between proofs (k_lookup_starts clears what it reads)
        if (fresh || S.lk_nkeys == 0) {
            // host staging lives in the circuit state (the copies below are asynchronous and nothing here waits)
            static_assert(sizeof(F) == 32, "scalar field element = 8 words");
            S.lk_host_keys.resize((table_len + 1) * 8);
            S.lk_host_sorted.resize((table_len + 1) * 8);
            std::vector<uint32_t>& counts = S.lk_host_counts;
            std::vector<uint32_t>& order = S.lk_host_order;
            struct Span {   // a growable view of F over the word vectors
                std::vector<uint32_t>& w;
                size_t len;
                F& operator[](size_t i) { return *reinterpret_cast<F*>(w.data() + 8 * i); }
                void resize(size_t k) { len = k; }
                void push_back(const F& v) { (*this)[len] = v; ++len; }
                size_t size() const { return len; }
                F* data() { return reinterpret_cast<F*>(w.data()); }
            };
            Span keys{S.lk_host_keys, 0}, sorted{S.lk_host_sorted, 0};
            keys.resize(table_len);
            for (size_t i = 0; i < table_len; ++i) keys[i] = H::from_words(table + 4 * i);
            counts.assign(table_len, 1u);
            // t is padded with zeros to n (lookup/table.rs:52-61): they join the zero key or create it
            size_t zero_idx = table_len;
            for (size_t i = 0; i < table_len; ++i)
                if (fe_is_zero<R>(keys[i])) {
                    zero_idx = i;
                    break;
                }
            if (n > table_len) {
                if (zero_idx == table_len) {
                    keys.push_back(fe_zero<R>());
                    counts.push_back(0);
                }
                counts[zero_idx] += (uint32_t)(n - table_len);
            }
            const uint32_t nk = (uint32_t)keys.size();
            if (nk + 2 > S.lk_cap) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "lookup table larger than the circuit bound");
            order.resize(nk);
            for (uint32_t i = 0; i < nk; ++i) order[i] = i;
            auto less = [&](uint32_t a, uint32_t b) {
                for (int i = R::N - 1; i >= 0; --i) {
                    if (keys[a].v[i] != keys[b].v[i]) return keys[a].v[i] < keys[b].v[i];
                }
                return false;
            };
            std::sort(order.begin(), order.end(), less);
            sorted.resize(nk);
            for (uint32_t i = 0; i < nk; ++i) sorted[i] = keys[order[i]];
            ZKT_HIP(c, hipMemcpyAsync(d_keys, keys.data(), nk * 32, hipMemcpyHostToDevice, c->stream));
            ZKT_HIP(c, hipMemcpyAsync(d_sorted, sorted.data(), nk * 32, hipMemcpyHostToDevice, c->stream));
            ZKT_HIP(c, hipMemcpyAsync(d_perm, order.data(), nk * 4, hipMemcpyHostToDevice, c->stream));
            ZKT_HIP(c, hipMemcpyAsync(d_base, counts.data(), nk * 4, hipMemcpyHostToDevice, c->stream));
            ZKT_HIP(c, hipMemsetAsync(d_hits, 0, (size_t)nk * 4, c->stream));   // a failed proof may have left counts behind
            S.lk_nkeys = nk;
        }
        const uint32_t nk = S.lk_nkeys;
        int rc;
        // a looked-up value outside the table sets status bit 4 (ElementNotIndexedInTable); like the other status
        // bits it is read at the next host round trip
        if ((rc = lookup_count(c, S.ev[4], n, d_sorted, d_perm, nk, d_hits, S.status))) return rc;
        if ((rc = lookup_starts(c, d_base, d_hits, nk, n, d_even, d_odd, S.status))) return rc;   // multiset.rs:126-143
        if ((rc = lookup_expand(c, d_keys, d_even, nk, S.ev[5], n))) return rc;
        if ((rc = lookup_expand(c, d_keys, d_odd, nk, S.ev[6], n))) return rc;
        return ZKT_OK;
    }

    static bool same_inputs(const zkt_prove_inputs& a, const zkt_prove_inputs& b) {
        return a.a_evals == b.a_evals && a.b_evals == b.b_evals && a.c_evals == b.c_evals && a.n_rows == b.n_rows &&
               a.table == b.table && a.table_len == b.table_len && a.pi_pos == b.pi_pos && a.pi_vals == b.pi_vals &&
               a.n_pi == b.n_pi && a.blinders == b.blinders && a.wires_on_device == b.wires_on_device &&
               a.variables == b.variables && a.n_vars == b.n_vars && a.w_l == b.w_l && a.w_r == b.w_r && a.w_o == b.w_o;
    }
    int to_coset(int k) {
        // quotient_poly.rs:52-96 needs every witness polynomial on the 4n coset.  None of those transforms depends on a
        // challenge, so each is issued right behind the commitment of its polynomial, where it hides the latency-bound
        // tail of the last MSM of the round (which runs on the side stream).
        static const int coset_src[W_COUNT] = {0, 1, 2, 8, 6, 7, 3, 4, 5};  // a b c pi z1 z2 t h1 h2
        if (S.G == 1) return ntt_run(c, S.log_n + 2, 0, 1, S.poly[coset_src[k]], S.n + 8, S.wcos[k]);
        // sharded proof: only this GPU's class of the 4n coset, one m-point transform, no exchange
        int rc = ntt_run_class(c, S.log_m, S.log_n + 2, S.cls, S.poly[coset_src[k]], S.n + 8, S.wcos[k], S.fold);
        if (rc || S.G < 8) return rc;
        static const int next_of[W_COUNT] = {-1, -1, -1, -1, 0, 1, 2, 3, -1};     // z1 z2 t h1 are read at "omega-next"
        if (next_of[k] < 0) return ZKT_OK;
        return ntt_run_class(c, S.log_m, S.log_n + 2, (S.cls + 4) & 7, S.poly[coset_src[k]], S.n + 8, S.wnext[next_of[k]], S.fold);
    }

    // several witness polynomials onto the 4n coset in one launch per pass (single GPU; a sharded proof transforms its
    // classes one by one)
    int to_coset_many(std::initializer_list<int> ks) {
        static const int coset_src[W_COUNT] = {0, 1, 2, 8, 6, 7, 3, 4, 5};  // a b c pi z1 z2 t h1 h2
        int rc;
        if (S.G != 1 || ks.size() > (size_t)NTT_MAX_BATCH) {
            for (int k : ks) if ((rc = to_coset(k))) return rc;
            return ZKT_OK;
        }
        const void* ins[NTT_MAX_BATCH];
        void* outs[NTT_MAX_BATCH];
        size_t lens[NTT_MAX_BATCH];
        int nb = 0;
        for (int k : ks) {
            ins[nb] = S.poly[coset_src[k]];
            outs[nb] = S.wcos[k];
            lens[nb] = S.n + 8;
            ++nb;
        }
        return nb ? ntt_run_batch(c, S.log_n + 2, 0, 1, nb, ins, lens, outs) : ZKT_OK;
    }

    void swap_work_sets() {   // current <-> alternate copies of what early work of the next proof overwrites
        for (int k = 0; k < 3; ++k) std::swap(S.poly[k], S.poly_alt[k]);
        std::swap(S.status, S.status_alt);
    }

    // Everything of rounds 1 and 2 that runs on the device (prove.rs:116-185): no challenge is needed before beta, so
    // these two parts touch no transcript and can be issued ahead of time (zkt_prove_set_next).  Commitments: slots 0-5.
    int enqueue_round_1(const zkt_prove_inputs& in) {
        const size_t n = S.n;
        int rc;
        ProfScope prof_round(c, "round1");
        if (in.n_rows > n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "more rows than the circuit bound");
        if (in.table_len >= n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "max table size is equal or larger than n");
        if (n < 8) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "circuit bound below 8");
        if (!in.blinders) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null blinders");
        ZKT_HIP(c, hipMemsetAsync(S.status, 0, 64 * 4, c->stream));
        ZKT_HIP(c, hipMemcpyAsync(S.small, in.blinders, 19 * 32, hipMemcpyHostToDevice, c->stream));

        // ---- round 1 (prove.rs:116-140) ----
        const uint64_t* wires[3] = {in.a_evals, in.b_evals, in.c_evals};
        const bool from_vars = in.a_evals == nullptr && in.variables != nullptr;
        const void* d_vars = in.variables;
        const uint32_t* d_idx[3] = {in.w_l, in.w_r, in.w_o};
        if (from_vars) {
            if (in.n_rows && (!in.w_l || !in.w_r || !in.w_o)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null wire index vector");
            if (in.n_vars > 0xFFFFFFFEull) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "too many variables");
            if (!in.wires_on_device) {   // stage the composer's arrays: values in the (still unused) quotient vector
                if (in.n_vars > 4 * n || 3 * in.n_rows * 4 > (n + 8) * 32)
                    return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "witness larger than the staging buffers");
                if (in.n_vars)
                    ZKT_HIP(c, hipMemcpyAsync(S.qev, in.variables, in.n_vars * 32, hipMemcpyHostToDevice, c->stream));
                d_vars = S.qev;
                for (int k = 0; k < 3; ++k) {
                    uint32_t* dst = (uint32_t*)S.poly[12] + (size_t)k * in.n_rows;   // the round-5 work buffer
                    if (in.n_rows)
                        ZKT_HIP(c, hipMemcpyAsync(dst, d_idx[k], in.n_rows * 4, hipMemcpyHostToDevice, c->stream));
                    d_idx[k] = dst;
                }
            }
        }
        const bool host_wires = !from_vars && !in.wires_on_device;
        if (host_wires) {
            // cold path: the three wire vectors cross PCIe inside the call.  They travel on the copy stream while the main
            // stream already transforms and commits the wire before, so only the first upload is exposed.
            if (!copy_fenced) {
                ZKT_HIP(c, hipEventRecord(S.ev_copy[3], c->stream));   // the targets are free once earlier work has drained
                ZKT_HIP(c, hipStreamWaitEvent(S.copy_stream, S.ev_copy[3], 0));
            }
            copy_fenced = false;
            for (int k = 0; k < 3; ++k) {
                if (in.n_rows < n)
                    ZKT_HIP(c, hipMemsetAsync((char*)S.ev[k] + in.n_rows * 32, 0, (n - in.n_rows) * 32, S.copy_stream));
                if (in.n_rows)
                    ZKT_HIP(c, hipMemcpyAsync(S.ev[k], wires[k], in.n_rows * 32, hipMemcpyHostToDevice, S.copy_stream));
                ZKT_HIP(c, hipEventRecord(S.ev_copy[k], S.copy_stream));
                ZKT_HIP(c, hipStreamWaitEvent(c->stream, S.ev_copy[k], 0));
                if ((rc = evals_to_blinded_poly(S.ev[k], S.poly[k], 2 * k, 2, k))) return rc;
                if ((rc = commit_begin(S.poly[k], n + 2, k))) return rc;
            }
        } else {
            for (int k = 0; k < 3; ++k) {
                if (from_vars) {
                    if ((rc = poly_gather_pad(c, d_vars, in.n_vars, d_idx[k], in.n_rows, S.ev[k], n, S.status))) return rc;
                } else {
                    if ((rc = poly_copy_pad(c, wires[k], in.n_rows, S.ev[k], n))) return rc;   // prove.rs:39-55 pad_to
                }
            }
            const PolyJob jobs[3] = {{S.ev[0], S.poly[0], 0, 2, 0}, {S.ev[1], S.poly[1], 2, 2, 1}, {S.ev[2], S.poly[2], 4, 2, 2}};
            if ((rc = evals_to_blinded_polys(jobs, 3))) return rc;
            {
                void* const polys[3] = {S.poly[0], S.poly[1], S.poly[2]};
                const size_t lens[3] = {n + 2, n + 2, n + 2};
                static const int slots[3] = {0, 1, 2};
                if ((rc = commit_begin_many(polys, lens, slots, 3))) return rc;
            }
        }
        if ((rc = commit_flush())) return rc;
        return to_coset_many({W_A, W_B, W_C});
    }

    // ---- round 2 (prove.rs:145-185): its three commitments join the batch of round 1
    // The table polynomial's part of round 2 (prove.rs:145-156,178-180): evaluations, coefficients, commitment, coset.
    // It depends on nothing but the lookup table, so a direct (unannounced) proof issues it FIRST: it then covers the
    // upload of the first wire vector of a cold proof.  Returns whether the resident table polynomial is reused.
    bool table_is_cached(const zkt_prove_inputs& in) const {
        // the cached commitment belongs to the SRS it was made under: a reloaded key invalidates it (the polynomial and
        // its coset would survive, but one flag keeps the three together)
        return S.t_cached && S.t_srs_generation == c->srs_generation && S.cached_table.size() == 4 * in.table_len &&
               (in.table_len == 0 || memcmp(S.cached_table.data(), in.table, in.table_len * 32) == 0);
    }
    int enqueue_table(const zkt_prove_inputs& in, bool with_coset) {
        const size_t n = S.n;
        int rc;
        if (in.table_len >= n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "max table size is equal or larger than n");
        S.t_cached = false;
        S.t_coset_valid = false;
        ZKT_HIP(c, hipMemsetAsync(S.ev[3], 0, n * 32, c->stream));
        if (in.table_len) ZKT_HIP(c, hipMemcpyAsync(S.ev[3], in.table, in.table_len * 32, hipMemcpyHostToDevice, c->stream));
        if ((rc = evals_to_blinded_poly(S.ev[3], S.poly[3], 0, 0, 3))) return rc;      // t: no blinders
        if ((rc = commit_evals_begin(S.ev[3], S.poly[3], 0, 0, 3, 3))) return rc;
        if (with_coset && (rc = to_coset(W_T))) return rc;
        return ZKT_OK;
    }

    // table_done: enqueue_table has already run for this proof (direct path)
    int enqueue_round_2(const zkt_prove_inputs& in, bool* same_table_out, bool table_done = false) {
        const size_t n = S.n;
        int rc;
        ProfScope prof_round(c, "round2");
        const bool same_table = !table_done && table_is_cached(in);
        *same_table_out = same_table;
        if (!same_table && !table_done && (rc = enqueue_table(in, false))) return rc;
        if ((rc = poly_mul_vec(c, S.q_lookup_ev, S.ev[2], S.ev[4], n))) return rc;         // f = q_lookup . c
        if ((rc = combine_split(in.table, in.table_len, !same_table))) return rc;
        {
            const PolyJob jobs[2] = {{S.ev[5], S.poly[4], 6, 3, 4}, {S.ev[6], S.poly[5], 9, 2, 5}};   // h1: 3 blinders, h2: 2
            if ((rc = evals_to_blinded_polys(jobs, 2))) return rc;
        }
        if ((rc = commit_evals_begin(S.ev[5], S.poly[4], 6, 3, 4, 4))) return rc;
        if ((rc = commit_evals_begin(S.ev[6], S.poly[5], 9, 2, 5, 5))) return rc;
        if ((rc = commit_flush())) return rc;
        if (!S.t_coset_valid && !table_done) return to_coset_many({W_T, W_H1, W_H2});
        return to_coset_many({W_H1, W_H2});   // unchanged table: its coset is still resident
    }

    int run(const zkt_prove_inputs& in, std::vector<uint8_t>& proof) {
        const size_t n = S.n;
        const int log_n = S.log_n;
        int rc;
        mark("start");
        bool same_table = false;
        if (S.prefetch_stage == 2 && S.prefetch_epoch == c->msm_epoch && same_inputs(S.prefetch_in, in)) {
            same_table = S.prefetch_same_table;   // rounds 1 and 2 are already in flight, in the alternate work set
            swap_work_sets();
            S.prefetch_stage = 0;
        } else {
            S.prefetch_stage = 0;
            // nobody announced this proof: a fresh lookup table's polynomial goes first (it waits for nothing, and its
            // commitment covers the first wire upload of a cold proof)
            const bool table_first = !table_is_cached(in);
            if (table_first && in.a_evals && !in.wires_on_device) {
                // host wire vectors: the copy stream is fenced HERE, so that the uploads run beside the table's work
                // instead of behind it
                ZKT_HIP(c, hipEventRecord(S.ev_copy[3], c->stream));
                ZKT_HIP(c, hipStreamWaitEvent(S.copy_stream, S.ev_copy[3], 0));
                copy_fenced = true;
            }
            if (table_first && (rc = enqueue_table(in, true))) return rc;
            if ((rc = enqueue_round_1(in))) return rc;
            if ((rc = enqueue_round_2(in, &same_table, table_first))) return rc;
        }

        // prove.rs:110 -- public inputs (BTreeMap order = ascending position)
        {
            std::vector<uint8_t> b(in.n_pi * 32);
            for (size_t i = 0; i < in.n_pi; ++i) H::to_le_bytes(H::from_words(in.pi_vals + 4 * i), b.data() + 32 * i);
            tr.append_scalars("pi", b.data(), in.n_pi, 32, false);
        }
        Affine<Q> cm[11];
        mark("enqueue rounds 1+2");
        {
            static const int slots[6] = {0, 1, 2, 3, 4, 5};
            const bool skip[6] = {false, false, false, same_table, false, false};
            if ((rc = commit_collect(slots, skip, 6, cm))) return rc;
        }
        mark("wait commits of rounds 1+2");
        if (!same_table) {
            memcpy(S.t_commit_xy, cm[3].x.v, Q::N * 4);
            memcpy(S.t_commit_xy + Q::N / 2, cm[3].y.v, Q::N * 4);
            S.cached_table.assign(in.table, in.table + 4 * in.table_len);
            S.t_cached = true;
            S.t_srs_generation = c->srs_generation;
        } else {
            memcpy(cm[3].x.v, S.t_commit_xy, Q::N * 4);
            memcpy(cm[3].y.v, S.t_commit_xy + Q::N / 2, Q::N * 4);
        }
        static const char* L12[6] = {"a_commit", "b_commit", "c_commit", "t_commit", "h1_commit", "h2_commit"};
        for (int k = 0; k < 6; ++k) tr_commit(L12[k], cm[k]);
        mark("transcript rounds 1+2");

        // ---- round 3 (prove.rs:190-255) ----
        const F beta = tr_challenge("beta"), gamma = tr_challenge("gamma"), delta = tr_challenge("delta"),
                epsilon = tr_challenge("epsilon");
        if (fe_eq<R>(beta, gamma) || fe_eq<R>(beta, delta) || fe_eq<R>(beta, epsilon) || fe_eq<R>(gamma, delta) ||
            fe_eq<R>(gamma, epsilon) || fe_eq<R>(delta, epsilon))
            return set_err(c, ZKT_ERR_EQUAL_CHALLENGES, "challenges must be different (prove.rs:202-207)");
        ZTermsArgs za{};
        za.a = S.ev[0]; za.b = S.ev[1]; za.c = S.ev[2];
        za.s1 = S.sigma_ev[0]; za.s2 = S.sigma_ev[1]; za.s3 = S.sigma_ev[2]; za.roots = S.roots;
        za.f = S.ev[4]; za.t = S.ev[3]; za.h1 = S.ev[5]; za.h2 = S.ev[6];
        za.num = S.sc[0]; za.den = S.sc[1]; za.n = n;
        put(za.beta, beta); put(za.gamma, gamma); put(za.delta, delta); put(za.epsilon, epsilon);
        F* pin = (F*)S.pinned;
        // Both grand products are prefix products of num/den ratios; the division is one inversion of the total
        // denominator on the host.  The two products are independent, so both pairs of scans run before the single
        // host round trip (the second pair borrows the still unused z1 coset buffer).
        void* pn2 = S.wcos[W_Z1];
        void* sd2 = (char*)S.wcos[W_Z1] + n * 32;
        mark("challenges round 3");
        std::optional<ProfScope> prof_round;   // stream time of a round, first launch to last (zkt_profile_get "round3" ...)
        wait_end();
        prof_round.emplace(c, "round3");
        if ((rc = z1_terms(c, za))) return rc;
        if ((rc = scan_mul(c, S.sc[0], S.sc[2], n, false, S.scan_tmp))) return rc;   // PN
        if ((rc = scan_mul(c, S.sc[1], S.sc[3], n, true, S.scan_tmp))) return rc;    // SD
        if ((rc = z2_terms(c, za))) return rc;
        if ((rc = scan_mul(c, S.sc[0], pn2, n, false, S.scan_tmp))) return rc;
        if ((rc = scan_mul(c, S.sc[1], sd2, n, true, S.scan_tmp))) return rc;
        ZKT_HIP(c, hipMemcpyAsync(pin, S.sc[3], 32, hipMemcpyDeviceToHost, c->stream));
        ZKT_HIP(c, hipMemcpyAsync(pin + 1, sd2, 32, hipMemcpyDeviceToHost, c->stream));
        mark("enqueue grand products");
        if ((rc = check_status())) return rc;   // synchronises; reports a lookup outside the table (round 2)
        mark("wait grand products");
        if (fe_is_zero<R>(pin[0])) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "zero denominator in the permutation grand product");
        if (fe_is_zero<R>(pin[1])) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "zero denominator in the lookup grand product");
        {
            const F d1 = pin[0], d2 = pin[1];
            const F inv12 = fe_inv_host<R>(fe_mul<R>(d1, d2));
            const F inv1 = fe_mul<R>(inv12, d2), inv2 = fe_mul<R>(inv12, d1);
            wait_end();
            if ((rc = z_combine(c, S.sc[2], S.sc[3], inv1.v, S.ev[7], n))) return rc;
            if ((rc = z_combine(c, pn2, sd2, inv2.v, S.sc[0], n))) return rc;              // num / den are free again
            const PolyJob jobs[2] = {{S.ev[7], S.poly[6], 11, 3, 6}, {S.sc[0], S.poly[7], 14, 3, 7}};   // z1, z2: 3 blinders each
            if ((rc = evals_to_blinded_polys(jobs, 2))) return rc;
        }
        {
            void* const z1p[1] = {S.poly[6]};
            const size_t z1l[1] = {n + 3};
            static const int z1s[1] = {0};
            if ((rc = commit_begin_many(z1p, z1l, z1s, 1))) return rc;
        }
        if ((rc = commit_evals_begin(S.sc[0], S.poly[7], 14, 3, 7, 1))) return rc;
        if ((rc = commit_flush())) return rc;
        if ((rc = to_coset_many({W_Z1, W_Z2}))) return rc;
        // the public-input polynomial of round 4 (prove.rs:258-262) is challenge-free as well.  With a handful of
        // public inputs it is never built: the quotient kernel evaluates it from rotations of l1 (poly.hpp).
        for (size_t i = 0; i < in.n_pi; ++i)
            if (in.pi_pos[i] >= n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "public input position out of range");
        // (with 8 GPUs a rotation by 4 pos leaves the class for odd pos: the polynomial is transformed instead)
        const bool pi_direct = in.n_pi <= (size_t)QUOTIENT_PI_DIRECT_MAX && S.G <= 4;
        if (pi_direct) {
            uint32_t* tab = (uint32_t*)S.pinned_pi;
            for (size_t i = 0; i < in.n_pi; ++i) {
                tab[10 * i] = (uint32_t)(4 * in.pi_pos[i] / (size_t)S.G);   // rotation in entries of the class
                const Fx<R> v = fx_unpack<R>(H::from_words(in.pi_vals + 4 * i));
                for (int w = 0; w < 9; ++w) tab[10 * i + 1 + w] = v.l[w];
            }
            if (in.n_pi)
                ZKT_HIP(c, hipMemcpyAsync(S.pi_tab, tab, in.n_pi * 40, hipMemcpyHostToDevice, c->stream));
        } else {
            ZKT_HIP(c, hipMemsetAsync(S.ev[7], 0, n * 32, c->stream));
            for (size_t i = 0; i < in.n_pi; ++i)
                ZKT_HIP(c, hipMemcpyAsync((char*)S.ev[7] + in.pi_pos[i] * 32, in.pi_vals + 4 * i, 32, hipMemcpyHostToDevice, c->stream));
            if ((rc = evals_to_blinded_poly(S.ev[7], S.poly[8], 0, 0, 8))) return rc;      // pi
            if ((rc = to_coset(W_PI))) return rc;
        }
        {
            static const int slots[2] = {0, 1};
            if ((rc = commit_collect(slots, nullptr, 2, cm + 6))) return rc;
        }
        prof_round.reset();
        tr_commit("z1_commit", cm[6]);
        tr_commit("z2_commit", cm[7]);
        mark("round 3 finish + commits");

        // ---- round 4 (prove.rs:258-313) ----
        const F alpha = tr_challenge("alpha");
        wait_end();
        prof_round.emplace(c, "round4");
        {
            S.t_coset_valid = S.t_cached;
            QuotientArgs q{};
            q.a = S.wcos[W_A]; q.b = S.wcos[W_B]; q.c = S.wcos[W_C]; q.pi = S.wcos[W_PI];
            q.z1 = S.wcos[W_Z1]; q.z2 = S.wcos[W_Z2]; q.t = S.wcos[W_T]; q.h1 = S.wcos[W_H1]; q.h2 = S.wcos[W_H2];
            q.q_m = S.coset[CS_QM]; q.q_l = S.coset[CS_QL]; q.q_r = S.coset[CS_QR]; q.q_o = S.coset[CS_QO];
            q.q_c = S.coset[CS_QC]; q.q_lookup = S.coset[CS_QLOOKUP]; q.q_table = S.coset[CS_QTABLE];
            q.sigma1 = S.coset[CS_S1]; q.sigma2 = S.coset[CS_S2]; q.sigma3 = S.coset[CS_S3];
            q.x = S.coset[CS_X]; q.l1 = S.coset[CS_L1];
            q.out = S.qev;
            put(q.alpha, alpha); put(q.beta, beta); put(q.gamma, gamma); put(q.delta, delta); put(q.epsilon, epsilon);
            memcpy(q.zh_inv, S.zh_inv, sizeof(q.zh_inv));
            q.n4 = S.m;
            q.pi_tab = pi_direct ? S.pi_tab : nullptr;
            q.n_pi_direct = pi_direct ? (uint32_t)in.n_pi : 0;
            if (S.G > 1) {
                // this GPU's class of the coset, written where the all-gather wants it; then the one real exchange of a
                // sharded proof: 4n x 32 B in total, and the classes go back to natural order for the inverse transform
                q.G = (uint32_t)S.G;
                q.cls = (uint32_t)S.cls;
                q.next_off = S.G <= 4 ? 4u / (uint32_t)S.G : (uint32_t)((S.cls + 4) >> 3);
                q.z1_next = S.G == 8 ? S.wnext[0] : q.z1;
                q.z2_next = S.G == 8 ? S.wnext[1] : q.z2;
                q.t_next = S.G == 8 ? S.wnext[2] : q.t;
                q.h1_next = S.G == 8 ? S.wnext[3] : q.h1;
                const size_t pieces = ZKT_QUOTIENT_CHUNKS;
                if (comm_async_available(c) && S.comm_stream && S.m % pieces == 0 && S.m / pieces >= 1024) {
                    // the exchange in pieces, stream-ordered, on a stream of its own: piece j's collective travels while
                    // piece j + 1 is computed, and the host never waits (zkt_comm_vtable::all_gather_async).  The gathered
                    // layout is [piece][class][m / pieces]; this rank's share of a piece is written where RCCL's in-place
                    // form wants it (recv + rank * bytes).
                    const size_t mp = S.m / pieces;
                    for (size_t j = 0; j < pieces; ++j) {
                        char* recv = (char*)S.qgather + j * (size_t)S.G * mp * 32;
                        char* mine = recv + (size_t)S.cls * mp * 32;
                        q.first = j * mp;
                        q.count = mp;
                        q.out = mine - q.first * 32;     // out[i] for i in the piece lands in `mine`
                        if ((rc = quotient_pointwise(c, q))) return rc;
                        ZKT_HIP(c, hipEventRecord(S.ev_piece[j], c->stream));
                        ZKT_HIP(c, hipStreamWaitEvent(S.comm_stream, S.ev_piece[j], 0));
                        if ((rc = comm_all_gather_async(c, mine, recv, mp * 32, S.comm_stream))) return rc;
                    }
                    ZKT_HIP(c, hipEventRecord(S.ev_gathered, S.comm_stream));
                    ZKT_HIP(c, hipStreamWaitEvent(c->stream, S.ev_gathered, 0));
                    if ((rc = quotient_interleave(c, S.qgather, S.qev, 4 * n, (uint32_t)S.G, (uint32_t)pieces))) return rc;
                } else {
                    q.out = (char*)S.qgather + (size_t)S.cls * S.m * 32;
                    if ((rc = quotient_pointwise(c, q))) return rc;
                    if ((rc = comm_all_gather_dev(c, q.out, S.qgather, S.m * 32))) return rc;
                    if ((rc = quotient_interleave(c, S.qgather, S.qev, 4 * n, (uint32_t)S.G))) return rc;
                }
            } else if ((rc = quotient_pointwise(c, q))) {
                return rc;
            }
            if ((rc = ntt_run(c, log_n + 2, 1, 1, S.qev, 4 * n, S.qev))) return rc;         // quotient_poly.rs:226
            if ((rc = quotient_split_blind(c, S.qev, n, (const char*)S.small + 17 * 32, S.poly[9], S.poly[10], S.poly[11], S.status)))
                return rc;
            // an unsatisfied circuit shows up as status bits here; they are read with the evaluations of round 5
        }
        {
            void* const polys[3] = {S.poly[9], S.poly[10], S.poly[11]};
            const size_t lens[3] = {n + 3, n + 3, n + 3};
            static const int slots[3] = {8, 9, 10};
            if ((rc = commit_begin_many(polys, lens, slots, 3))) return rc;
            if ((rc = commit_flush())) return rc;
        }
        if (S.has_next) {   // zkt_prove_set_next: round 1 of the next proof hides the tail of the quotient commitments
            S.has_next = false;
            swap_work_sets();
            const int r1 = enqueue_round_1(S.next_in);
            swap_work_sets();
            S.prefetch_stage = (r1 == ZKT_OK) ? 1 : 0;   // on failure the next zkt_prove redoes the work and reports
        }
        {
            static const int slots[3] = {8, 9, 10};
            if ((rc = commit_collect(slots, nullptr, 3, cm + 8))) return rc;
        }
        prof_round.reset();
        tr_commit("q_lo_commit", cm[8]);
        tr_commit("q_mid_commit", cm[9]);
        tr_commit("q_hi_commit", cm[10]);
        mark("round 4");

        // ---- round 5 (prove.rs:318-451, linearization_poly.rs:19-121) ----
        const F xi = tr_challenge("xi");
        wait_end();
        prof_round.emplace(c, "round5");
        F w;
        {
            Fe<R> g = root_of_unity<R>(log_n);
            w = g;
        }
        const F shifted = fe_mul<R>(xi, w);
        const size_t cap = n + 8;
        EvalArgs ea{};
        // order of ProofEvaluations: a b c | sigma1 sigma2 z1_next | q_lookup t t_next z2_next h1_next h2
        const void* ep[12] = {S.poly[0], S.poly[1], S.poly[2], S.pk[PK_S1], S.pk[PK_S2], S.poly[6],
                              S.pk[PK_QLOOKUP], S.poly[3], S.poly[3], S.poly[7], S.poly[4], S.poly[5]};
        const size_t el[12] = {cap, cap, cap, S.pk_len[PK_S1], S.pk_len[PK_S2], cap, S.pk_len[PK_QLOOKUP], cap, cap, cap, cap, cap};
        const bool sh[12] = {false, false, false, false, false, true, false, false, true, true, true, false};
        ea.count = 12;
        for (int k = 0; k < 12; ++k) {
            ea.poly[k] = ep[k];
            ea.len[k] = el[k];
            put(ea.point[k], sh[k] ? shifted : xi);
        }
        void* d_partials = (char*)S.small + 64 * 32;
        void* d_results = (char*)S.small + 32 * 32;
        if ((rc = poly_eval_many(c, ea, d_partials, d_results, S.eval_pw))) return rc;
        ZKT_HIP(c, hipMemcpyAsync(pin, d_results, 12 * 32, hipMemcpyDeviceToHost, c->stream));
        mark("enqueue evaluations");
        if ((rc = check_status())) return rc;   // synchronises the stream
        mark("wait evaluations");
        F ev[12];
        for (int k = 0; k < 12; ++k) ev[k] = pin[k];
        const F &e_a = ev[0], &e_b = ev[1], &e_c = ev[2], &e_s1 = ev[3], &e_s2 = ev[4], &e_z1n = ev[5], &e_ql = ev[6],
                &e_t = ev[7], &e_tn = ev[8], &e_z2n = ev[9], &e_h1n = ev[10], &e_h2 = ev[11];
        const F one = fe_one<R>();
        // zh(xi) = xi^n - 1 ; L_1(xi) = zh * 1 / (n * (xi - 1))   (util.rs:185-195)
        const F xi_n = fe_pow_u64<R>(xi, (uint64_t)n);
        const F zh = fe_sub<R>(xi_n, one);
        F nn = fe_zero<R>();
        nn.v[0] = (uint32_t)(n & 0xffffffffu);
        nn.v[1] = (uint32_t)((uint64_t)n >> 32);
        nn = fe_to_mont<R>(nn);
        const F l1den = fe_mul<R>(nn, fe_sub<R>(xi, one));
        if (fe_is_zero<R>(l1den)) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "xi = 1");
        const F l1 = fe_mul<R>(zh, fe_inv_host<R>(l1den));
        const F a2 = fe_sqr<R>(alpha), a3 = fe_mul<R>(a2, alpha), a4 = fe_mul<R>(a3, alpha), a5 = fe_mul<R>(a4, alpha);
        const F k1 = fe_from_u32<R>(7), k2 = fe_from_u32<R>(13);
        const F bxi = fe_mul<R>(beta, xi);
        // keys/permutation.rs:34-69
        F s_z1 = fe_mul<R>(alpha, fe_add<R>(fe_add<R>(bxi, e_a), gamma));
        s_z1 = fe_mul<R>(s_z1, fe_add<R>(fe_add<R>(fe_mul<R>(bxi, k1), e_b), gamma));
        s_z1 = fe_mul<R>(s_z1, fe_add<R>(fe_add<R>(fe_mul<R>(bxi, k2), e_c), gamma));
        s_z1 = fe_add<R>(s_z1, fe_mul<R>(l1, a2));
        F s_s3 = fe_mul<R>(fe_neg<R>(alpha), beta);
        s_s3 = fe_mul<R>(s_s3, e_z1n);
        s_s3 = fe_mul<R>(s_s3, fe_add<R>(fe_add<R>(fe_mul<R>(beta, e_s1), e_a), gamma));
        s_s3 = fe_mul<R>(s_s3, fe_add<R>(fe_add<R>(fe_mul<R>(beta, e_s2), e_b), gamma));
        // keys/lookup.rs:29-65
        const F opd = fe_add<R>(delta, one), eopd = fe_mul<R>(epsilon, opd);
        F s_z2 = fe_mul<R>(a3, opd);
        s_z2 = fe_mul<R>(s_z2, fe_add<R>(epsilon, fe_mul<R>(e_ql, e_c)));
        s_z2 = fe_mul<R>(s_z2, fe_add<R>(fe_add<R>(eopd, e_t), fe_mul<R>(delta, e_tn)));
        s_z2 = fe_add<R>(s_z2, fe_mul<R>(a4, l1));
        F s_h1 = fe_mul<R>(fe_neg<R>(a3), e_z2n);
        s_h1 = fe_mul<R>(s_h1, fe_add<R>(fe_add<R>(eopd, e_h2), fe_mul<R>(delta, e_h1n)));
        const F s_qt = fe_mul<R>(a5, e_t);
        // linearization_poly.rs:100-109: -zh * (q_lo + xi^(n+2) q_mid + xi^(2n+4) q_hi)
        const F xn2 = fe_mul<R>(fe_mul<R>(fe_add<R>(zh, one), xi), xi);
        const F nzh = fe_neg<R>(zh);
        LinCombArgs lr{};
        auto term = [&](LinCombArgs& L, const void* p, size_t len, const F& s) {
            L.poly[L.nterms] = p;
            L.len[L.nterms] = len;
            put(L.scalar[L.nterms], s);
            ++L.nterms;
        };
        term(lr, S.pk[PK_QM], S.pk_len[PK_QM], fe_mul<R>(e_a, e_b));   // keys/arithmetic.rs:37-46
        term(lr, S.pk[PK_QL], S.pk_len[PK_QL], e_a);
        term(lr, S.pk[PK_QR], S.pk_len[PK_QR], e_b);
        term(lr, S.pk[PK_QO], S.pk_len[PK_QO], e_c);
        term(lr, S.pk[PK_QC], S.pk_len[PK_QC], one);
        term(lr, S.poly[6], cap, s_z1);
        term(lr, S.pk[PK_S3], S.pk_len[PK_S3], s_s3);
        term(lr, S.poly[7], cap, s_z2);
        term(lr, S.poly[4], cap, s_h1);
        term(lr, S.pk[PK_QTABLE], S.pk_len[PK_QTABLE], s_qt);
        term(lr, S.poly[9], cap, nzh);
        term(lr, S.poly[10], cap, fe_mul<R>(nzh, xn2));
        term(lr, S.poly[11], cap, fe_mul<R>(nzh, fe_sqr<R>(xn2)));
        static const char* EL[12] = {"a_eval", "b_eval", "c_eval", "sigma1_eval", "sigma2_eval", "z1_next_eval",
                                     "q_lookup_eval", "t_eval", "t_next_eval", "z2_next_eval", "h1_next_eval", "h2_eval"};
        for (int k = 0; k < 12; ++k) tr_scalar(EL[k], ev[k]);
        const F eta = tr_challenge("eta");
        // r(X) lives in the work buffer; its own commitment (prove.rs:372-375) is never part of the proof
        // or the transcript, so it is not computed.
        // aw opening (prove.rs:381-420): sum_k eta^k p_k over (r, a, b, c, sigma1, sigma2, q_lookup, t, h2)
        // = eta^0 * r + ... : fold r's 13 terms and the 8 others into two passes
        void* work = S.poly[12];
        wait_end();
        if ((rc = poly_lincomb(c, lr, work, cap))) return rc;
        if (fe_is_zero<R>(xi) || fe_is_zero<R>(shifted)) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "evaluation challenge is zero");
        Affine<Q> aw, saw;
        {
            LinCombArgs lo{};
            F pw = one;
            term(lo, work, cap, pw);
            const void* ps[8] = {S.poly[0], S.poly[1], S.poly[2], S.pk[PK_S1], S.pk[PK_S2], S.pk[PK_QLOOKUP], S.poly[3], S.poly[5]};
            const size_t pl[8] = {cap, cap, cap, S.pk_len[PK_S1], S.pk_len[PK_S2], S.pk_len[PK_QLOOKUP], cap, cap};
            for (int k = 0; k < 8; ++k) {
                pw = fe_mul<R>(pw, eta);
                term(lo, ps[k], pl[k], pw);
            }
            void* comb = S.sc[0];  // n + 8 fits: sc buffers hold n + 8 elements
            if ((rc = poly_lincomb(c, lo, comb, cap))) return rc;
            F zi = fe_inv_host<R>(xi);
            if ((rc = open_witness(c, comb, cap, xi.v, zi.v, S.sc[1], S.sc[2], S.scan_tmp, S.sc[3], S.eval_pw))) return rc;
        }
        // saw opening (prove.rs:427-451): (z1, z2, t, h1) at xi * omega.  Its linear combination and division depend on
        // nothing the first opening computes: on a single GPU they run on a second stream beside the first opening's
        // commitment (memory-bound work beside an integer-ALU-bound kernel), in the z cosets' buffers, which are dead since
        // the quotient pass.
        const bool aux = S.aux_stream != nullptr && !c->aux_off;
        void* comb2 = aux ? S.wcos[W_Z1] : S.sc[0];
        void* ta2 = aux ? (void*)((char*)S.wcos[W_Z1] + cap * 32) : S.sc[1];
        void* tb2 = aux ? (void*)((char*)S.wcos[W_Z1] + 2 * cap * 32) : S.sc[2];
        void* out2 = aux ? S.wcos[W_Z2] : S.sc[3];
        auto second_opening = [&]() -> int {
            LinCombArgs lo{};
            F pw = one;
            const void* ps[4] = {S.poly[6], S.poly[7], S.poly[3], S.poly[4]};
            for (int k = 0; k < 4; ++k) {
                term(lo, ps[k], cap, pw);
                pw = fe_mul<R>(pw, eta);
            }
            int r2;
            if ((r2 = poly_lincomb(c, lo, comb2, cap))) return r2;
            F zi = fe_inv_host<R>(shifted);
            return open_witness(c, comb2, cap, shifted.v, zi.v, ta2, tb2, aux ? S.aux_scan_tmp : S.scan_tmp, out2,
                                aux ? S.aux_pw : S.eval_pw);
        };
        if (aux) {
            ZKT_HIP(c, hipEventRecord(S.ev_aux_go, c->stream));
            ZKT_HIP(c, hipStreamWaitEvent(S.aux_stream, S.ev_aux_go, 0));
            hipStream_t main_stream = c->stream;
            c->stream = S.aux_stream;           // the polynomial kernels launch on the context's stream
            const int r2 = second_opening();
            c->stream = main_stream;
            if (r2) return r2;
            ZKT_HIP(c, hipEventRecord(S.ev_aux_done, S.aux_stream));
        }
        // (small and mid-size keys: both commitments as one batch, which the second witness' own buffers allow)
        const bool both = aux && queueing();
        if (!both && (rc = commit_begin(S.sc[3], cap - 1, 6))) return rc;  // the scalars are consumed by the first kernels
        {
            if (aux) ZKT_HIP(c, hipStreamWaitEvent(c->stream, S.ev_aux_done, 0));
            else if ((rc = second_opening())) return rc;
            if (both) {
                void* const wp[2] = {S.sc[3], out2};
                const size_t wl[2] = {cap - 1, cap - 1};
                static const int ws[2] = {6, 7};
                if ((rc = commit_begin_many(wp, wl, ws, 2))) return rc;
            } else if ((rc = commit_begin(out2, cap - 1, 7))) {
                return rc;
            }
            if ((rc = commit_flush())) return rc;
            if (S.prefetch_stage == 1) {   // ... and its round 2 keeps the GPU fed across the proof boundary
                bool st = false;
                swap_work_sets();
                const int r2 = enqueue_round_2(S.next_in, &st);
                swap_work_sets();
                if (r2 == ZKT_OK) {
                    S.prefetch_stage = 2;
                    S.prefetch_in = S.next_in;
                    S.prefetch_same_table = st;
                    S.prefetch_epoch = c->msm_epoch;   // any other MSM before the next proof invalidates it
                } else {
                    S.prefetch_stage = 0;
                }
            }
            {
                static const int slots[2] = {6, 7};
                Affine<Q> w2[2];
                if ((rc = commit_collect(slots, nullptr, 2, w2))) return rc;
                aw = w2[0];
                saw = w2[1];
            }
        }
        wait_end();
        prof_round.reset();

        // ---- Proof (proof.rs:106-155), CanonicalSerialize ----
        proof.clear();
        for (int k = 0; k < 11; ++k) serialize_point<Q>(cm[k], proof);
        serialize_point<Q>(aw, proof);
        proof.push_back(0);   // kzg10::Proof::random_v = None
        serialize_point<Q>(saw, proof);
        proof.push_back(0);
        for (int k = 0; k < 12; ++k) {
            uint8_t b[32];
            H::to_le_bytes(ev[k], b);
            proof.insert(proof.end(), b, b + 32);
        }
        mark("round 5 finish");
        dump_marks();
        return ZKT_OK;
    }
};

// ---- circuit (ProverKey + ExtendedProverKey) ------------------------------------------------------
void circuit_release(zkt_ctx* c) {
    if (!c->circuit) return;
    (void)hipStreamSynchronize(c->stream);
    CircuitState& S = *c->circuit;
    if (S.copy_stream) (void)hipStreamSynchronize(S.copy_stream);
    auto fr = [&](void* p) { dev_free(c, p); };
    if (!S.keys_borrowed) {
        for (void* p : S.pk) fr(p);
        for (void* p : S.coset) fr(p);
        for (void* p : S.sigma_ev) fr(p);
        fr(S.q_lookup_ev); fr(S.roots);
    }
    for (void* p : S.ev) fr(p);
    for (void* p : S.sc) fr(p);
    fr(S.scan_tmp);
    for (void* p : S.poly) fr(p);
    for (void* p : S.wcos) fr(p);
    for (void* p : S.wnext) fr(p);
    fr(S.fold); fr(S.qgather);
    for (void* p : S.poly_alt) fr(p);
    fr(S.status_alt);
    fr(S.aux_scan_tmp); fr(S.aux_pw);
    for (void* q : S.lag_scalars_q) fr(q);
    if (S.aux_stream) {
        (void)hipStreamSynchronize(S.aux_stream);
        (void)hipStreamDestroy(S.aux_stream);
    }
    if (S.ev_aux_go) (void)hipEventDestroy(S.ev_aux_go);
    if (S.ev_aux_done) (void)hipEventDestroy(S.ev_aux_done);
    fr(S.qev); fr(S.small); fr(S.lag_scalars); fr(S.status); fr(S.lk_u32); fr(S.lk_keys); fr(S.pi_tab); fr(S.eval_pw);
    if (S.pinned) (void)hipHostFree(S.pinned);
    if (S.pinned_pi) (void)hipHostFree(S.pinned_pi);
    if (S.copy_stream) {
        (void)hipStreamSynchronize(S.copy_stream);
        (void)hipStreamDestroy(S.copy_stream);
    }
    for (auto e : S.ev_copy) if (e) (void)hipEventDestroy(e);
    if (S.comm_stream) {
        (void)hipStreamSynchronize(S.comm_stream);
        (void)hipStreamDestroy(S.comm_stream);
    }
    for (auto e : S.ev_piece) if (e) (void)hipEventDestroy(e);
    if (S.ev_gathered) (void)hipEventDestroy(S.ev_gathered);
    c->circuit.reset();
}

// Everything a context needs PER PROOF (work buffers, staging, streams, events) for the circuit shape in S (log_n, G, cls
// set): what a forked context allocates for itself while the keys stay its parent's.
static int circuit_alloc_work(zkt_ctx* c, CircuitState& S) {
    const size_t n = S.n, m = S.m;
    int rc;
    auto alloc = [&](void** p, size_t elems) { return dev_alloc(c, p, elems * 32); };
    if ((rc = alloc(&S.qev, 4 * n))) return rc;
    if (S.G > 1) {
        if ((rc = alloc(&S.fold, m))) return rc;
        if ((rc = alloc(&S.qgather, 4 * n))) return rc;
        if (S.G == 8) for (auto& p : S.wnext) if ((rc = alloc(&p, m))) return rc;
    }
    for (auto& p : S.ev) if ((rc = alloc(&p, n + 8))) return rc;
    for (auto& p : S.sc) if ((rc = alloc(&p, n + 8))) return rc;
    if ((rc = alloc(&S.scan_tmp, 2 * ((n + 8) / 1024 + 4096)))) return rc;
    for (auto& p : S.poly) if ((rc = alloc(&p, n + 8))) return rc;
    for (int k = 0; k < W_COUNT; ++k)   // the z1 slot doubles as 2n elements of scan scratch in round 3
        if ((rc = alloc(&S.wcos[k], k == W_Z1 ? std::max(m, 2 * n) : m))) return rc;
    const size_t eval_blocks = (n + 8 + 2047) / 2048 + 1;
    if ((rc = alloc(&S.small, 64 + 16 * eval_blocks))) return rc;
    if ((rc = alloc(&S.lag_scalars, n + 16))) return rc;   // n + 8 scalars, then the blinders of zkt_commit_evals_dev
    for (auto& q : S.lag_scalars_q) if ((rc = alloc(&q, n + 16))) return rc;
    if ((rc = dev_alloc(c, (void**)&S.status, 64 * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&S.status_alt, 64 * 4))) return rc;
    for (auto& p : S.poly_alt) if ((rc = alloc(&p, n + 8))) return rc;
    S.lk_cap = n + 2;
    if ((rc = dev_alloc(c, (void**)&S.lk_u32, 5 * S.lk_cap * 4 + 16))) return rc;
    if ((rc = alloc(&S.lk_keys, 2 * S.lk_cap))) return rc;
    ZKT_HIP(c, hipHostMalloc(&S.pinned, 64 * 32));
    ZKT_HIP(c, hipStreamCreateWithFlags(&S.copy_stream, hipStreamNonBlocking));
    if (S.G > 1) {
        ZKT_HIP(c, hipStreamCreateWithFlags(&S.comm_stream, hipStreamNonBlocking));
        for (auto& e : S.ev_piece) ZKT_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ZKT_HIP(c, hipEventCreateWithFlags(&S.ev_gathered, hipEventDisableTiming));
    }
    for (auto& e : S.ev_copy) ZKT_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ZKT_HIP(c, hipHostMalloc(&S.pinned_pi, QUOTIENT_PI_DIRECT_MAX * 40));
    if ((rc = dev_alloc(c, (void**)&S.pi_tab, QUOTIENT_PI_DIRECT_MAX * 40))) return rc;
    if ((rc = alloc(&S.eval_pw, std::max((size_t)EVAL_MAX * (257 + eval_blocks), open_witness_powers(n + 8))))) return rc;
    if (S.G == 1) {
        if ((rc = alloc(&S.aux_scan_tmp, 2 * ((n + 8) / 1024 + 4096)))) return rc;
        if ((rc = alloc(&S.aux_pw, open_witness_powers(n + 8)))) return rc;
        ZKT_HIP(c, hipStreamCreateWithFlags(&S.aux_stream, hipStreamNonBlocking));
        ZKT_HIP(c, hipEventCreateWithFlags(&S.ev_aux_go, hipEventDisableTiming));
        ZKT_HIP(c, hipEventCreateWithFlags(&S.ev_aux_done, hipEventDisableTiming));
    }
    return ZKT_OK;
}

// zkt_ctx_fork: the parent's keys (ProverKey polynomials, ExtendedProverKey cosets, sigma / q_lookup evaluations, domain
// roots: all read-only for a prover), this context's own work set
int circuit_fork(zkt_ctx* child, const zkt_ctx* parent) {
    circuit_release(child);
    if (!parent->circuit) return ZKT_OK;
    const CircuitState& P0 = *parent->circuit;
    auto st = std::make_shared<CircuitState>();
    CircuitState& S = *st;
    S.log_n = P0.log_n; S.n = P0.n; S.G = P0.G; S.cls = P0.cls; S.log_m = P0.log_m; S.m = P0.m;
    for (int k = 0; k < PK_COUNT; ++k) { S.pk[k] = P0.pk[k]; S.pk_len[k] = P0.pk_len[k]; }
    for (int k = 0; k < CS_COUNT; ++k) S.coset[k] = P0.coset[k];
    for (int k = 0; k < 3; ++k) S.sigma_ev[k] = P0.sigma_ev[k];
    S.q_lookup_ev = P0.q_lookup_ev;
    S.roots = P0.roots;
    memcpy(S.zh_inv, P0.zh_inv, sizeof(S.zh_inv));
    S.keys_borrowed = true;
    child->circuit = st;                      // from here on circuit_release cleans up after a failure
    const int rc = circuit_alloc_work(child, S);
    if (rc) circuit_release(child);
    return rc;
}

template <class C>
static int circuit_load_t(zkt_ctx* c, int log_n, const uint64_t* const* polys, const size_t* lens, bool from_evals = false,
                          bool from_evals_on_device = false) {
    using R = typename C::Fr;
    using F = Fe<R>;
    if (log_n < 3 || log_n + 2 > R::TWO_ADICITY || log_n + 2 > 27)
        return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "InvalidEvalDomainSize: 4n domain unsupported");
    if (int rf = refuse_if_forked(c, "loading a circuit")) return rf;
    circuit_release(c);
    auto st = std::make_shared<CircuitState>();
    CircuitState& S = *st;
    S.log_n = log_n;
    const size_t n = (size_t)1 << log_n;
    S.n = n;
    // sharded proof: this GPU keeps (and later transforms) only its class of the 4n coset
    S.G = c->sharded() ? c->comm.vt.world : 1;
    S.cls = c->sharded() ? c->comm.vt.rank : 0;
    S.log_m = log_n + 2 - (S.G == 8 ? 3 : S.G == 4 ? 2 : S.G == 2 ? 1 : 0);
    S.m = (size_t)1 << S.log_m;
    const size_t m = S.m;
    int rc;
    auto alloc = [&](void** p, size_t elems) { return dev_alloc(c, p, elems * 32); };
    if ((rc = circuit_alloc_work(c, S))) return rc;   // (the quotient vector doubles as staging below)
    for (int k = 0; k < PK_COUNT; ++k) {
        if (lens[k] > n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "prover-key polynomial longer than n");
        if ((rc = alloc(&S.pk[k], n))) return rc;
        if (from_evals) {
            // setup.rs:72-90: poly_from_evals of the padded selector / sigma / table-mask evaluations
            const void* src = polys[k];
            if (!from_evals_on_device) {
                ZKT_HIP(c, hipMemsetAsync(S.qev, 0, n * 32, c->stream));   // the quotient vector doubles as staging
                if (lens[k]) ZKT_HIP(c, hipMemcpyAsync(S.qev, polys[k], lens[k] * 32, hipMemcpyHostToDevice, c->stream));
                src = S.qev;
            }
            if ((rc = ntt_run(c, log_n, 1, 0, src, from_evals_on_device ? lens[k] : n, S.pk[k]))) return rc;
            S.pk_len[k] = n;
        } else {
            ZKT_HIP(c, hipMemsetAsync(S.pk[k], 0, n * 32, c->stream));
            if (lens[k]) ZKT_HIP(c, hipMemcpyAsync(S.pk[k], polys[k], lens[k] * 32, hipMemcpyHostToDevice, c->stream));
            S.pk_len[k] = lens[k];
        }
    }
    for (int k = 0; k < CS_COUNT; ++k) if ((rc = alloc(&S.coset[k], m))) return rc;
    for (int k = 0; k < 3; ++k) if ((rc = alloc(&S.sigma_ev[k], n))) return rc;
    if ((rc = alloc(&S.q_lookup_ev, n))) return rc;
    if ((rc = alloc(&S.roots, n))) return rc;

    // extend_prover_key (keys/mod.rs:78-146) on the device
    const int pk_of_cs[10] = {PK_QM, PK_QL, PK_QR, PK_QO, PK_QC, PK_QLOOKUP, PK_QTABLE, PK_S1, PK_S2, PK_S3};
    auto key_coset = [&](const void* poly, size_t len, void* out) {   // keys/mod.rs:98-120 coset_fft, this GPU's class
        return S.G == 1 ? ntt_run(c, log_n + 2, 0, 1, poly, len, out)
                        : ntt_run_class(c, S.log_m, log_n + 2, S.cls, poly, len, out, S.fold);
    };
    for (int k = 0; k < 10; ++k)
        if ((rc = key_coset(S.pk[pk_of_cs[k]], n, S.coset[k]))) return rc;
    // sigma / q_lookup evaluations on the n domain (prove.rs:91-94)
    if ((rc = ntt_run(c, log_n, 0, 0, S.pk[PK_S1], n, S.sigma_ev[0]))) return rc;
    if ((rc = ntt_run(c, log_n, 0, 0, S.pk[PK_S2], n, S.sigma_ev[1]))) return rc;
    if ((rc = ntt_run(c, log_n, 0, 0, S.pk[PK_S3], n, S.sigma_ev[2]))) return rc;
    if ((rc = ntt_run(c, log_n, 0, 0, S.pk[PK_QLOOKUP], n, S.q_lookup_ev))) return rc;
    const F one = fe_one<R>();
    const F w = root_of_unity<R>(log_n), w4n = root_of_unity<R>(log_n + 2);
    const F g = fe_from_u32<R>(R::GENERATOR);
    if ((rc = gen_powers(c, S.roots, n, w.v, one.v))) return rc;                 // domain.elements()
    {   // x_coset = g * w4n^t (keys/mod.rs:110-113), t = cls + G i on this GPU
        const F step = fe_pow_u64<R>(w4n, (uint64_t)S.G), first = fe_mul<R>(g, fe_pow_u64<R>(w4n, (uint64_t)S.cls));
        if ((rc = gen_powers(c, S.coset[CS_X], m, step.v, first.v))) return rc;
    }
    {   // l_1_coset = coset_fft(L_1), L_1 = ifft(1, 0, ..., 0) = (1/n, ..., 1/n)  (keys/mod.rs:119-120)
        F nn = fe_zero<R>();
        nn.v[0] = (uint32_t)(n & 0xffffffffu);
        nn.v[1] = (uint32_t)((uint64_t)n >> 32);
        F ninv = fe_inv_host<R>(fe_to_mont<R>(nn));
        if ((rc = gen_powers(c, S.ev[0], n, one.v, ninv.v))) return rc;
        if ((rc = key_coset(S.ev[0], n, S.coset[CS_L1]))) return rc;
    }
    // the tables that only ever multiply go to the quotient kernel's own Montgomery radix (poly.hpp)
    if ((rc = to_hat_form(c, S.coset[CS_QM], m, 1))) return rc;
    for (int k : {CS_QL, CS_QR, CS_QO, CS_QLOOKUP, CS_QTABLE, CS_L1})
        if ((rc = to_hat_form(c, S.coset[k], m, 0))) return rc;
    {   // zh_coset takes four values: (g * w4n^i)^n - 1 = g^n * w4^(i mod 4) - 1   (keys/mod.rs:115-117)
        const F gn = fe_pow_u64<R>(g, (uint64_t)n);
        const F w4 = fe_pow_u64<R>(w4n, (uint64_t)n);
        F cur = gn;
        for (int j = 0; j < 4; ++j) {
            F zh = fe_sub<R>(cur, one);
            if (fe_is_zero<R>(zh)) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "vanishing polynomial is zero on the coset");
            F inv = fe_inv_host<R>(zh);
            memcpy(S.zh_inv[j], inv.v, 32);
            cur = fe_mul<R>(cur, w4);
        }
    }
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    c->circuit = st;
    return ZKT_OK;
}

// setup.rs:42-166 on the device: polynomials from the evaluation vectors, ExtendedProverKey, the ten commitments
template <class C>
static int circuit_setup_t(zkt_ctx* c, int log_n, const uint64_t* const* evals, const size_t* lens, int on_device,
                           uint64_t* out_commitments, int* out_is_inf) {
    using Q = typename C::Fq;
    int rc = circuit_load_t<C>(c, log_n, evals, lens, true, on_device != 0);
    if (rc) return rc;
    CircuitState& S = *c->circuit;
    // setup.rs:104-121: PC::commit of the ten labelled polynomials, ProverKey order; batches of the MSM's slot count
    const int L = Q::N / 2;
    size_t off = 0, cnt = 0, total = 0;
    msm_slice(c, &off, &cnt, &total);
    if (c->sharded() && S.n > total)
        return set_err(c, ZKT_ERR_TOO_MANY_COEFFICIENTS, "TooManyCoefficients: polynomial longer than the committer key");
    for (int k0 = 0; k0 < PK_COUNT; k0 += 5) {
        const int kk = std::min(5, PK_COUNT - k0);
        uint64_t xy[5 * 12] = {};
        if (!c->sharded()) {
            for (int k = 0; k < kk; ++k)
                if ((rc = msm_begin(c, S.pk[k0 + k], S.n, 0, 1, k))) return rc;
            for (int k = 0; k < kk; ++k)
                if ((rc = msm_end(c, k, xy + 12 * k))) return rc;
        } else {   // this rank's slice of the coefficients; one all-gather per batch
            int slots[5];
            bool have[5];
            for (int k = 0; k < kk; ++k) {
                slots[k] = k;
                have[k] = S.n > off;
                if (have[k] && (rc = msm_begin(c, (const char*)S.pk[k0 + k] + off * 32, std::min(S.n, off + cnt) - off, 0, 1, k)))
                    return rc;
            }
            if ((rc = msm_end_sharded(c, slots, have, kk, xy))) return rc;
        }
        for (int k = 0; k < kk; ++k) {
            bool inf = true;   // the identity comes back as (0, 0)
            for (int i = 0; i < 2 * L; ++i) inf = inf && xy[12 * k + i] == 0;
            if (out_is_inf) out_is_inf[k0 + k] = inf ? 1 : 0;
            for (int i = 0; i < 2 * L; ++i) out_commitments[(size_t)(k0 + k) * 2 * L + i] = xy[12 * k + i];
        }
    }
    return ZKT_OK;
}

struct VtableTranscript;  // capi glue below

}  // namespace zkt

using namespace zkt;

namespace zkt {
// Adapter over a foreign transcript (the Rust shim's `T: TranscriptProtocol`): values cross in
// arkworks' Montgomery limbs, exactly what the shim can hand to T::append_scalar / append_commitment.
template <class C>
struct VtableAdapter : HostTranscript {
    using R = typename C::Fr;
    using Q = typename C::Fq;
    const zkt_transcript_vtable* vt;
    explicit VtableAdapter(const zkt_transcript_vtable* v) : vt(v) {}
    void append_u64(const char* label, uint64_t v) override { vt->append_u64(vt->user, label, v); }
    void append_scalars(const char* label, const uint8_t* le, size_t count, size_t, bool single) override {
        std::vector<uint64_t> m(4 * count);
        for (size_t i = 0; i < count; ++i) {
            Fe<R> x = HostF<R>::from_le_bytes(le + 32 * i);
            memcpy(&m[4 * i], x.v, 32);
        }
        vt->append_scalars(vt->user, label, m.data(), count, single ? 1 : 0);
    }
    void append_commitment(const char* label, const uint8_t* x_le, const uint8_t* y_le, size_t, bool inf) override {
        uint64_t xy[12] = {0};
        if (!inf) {
            Fe<Q> x, y;
            memcpy(x.v, x_le, Q::N * 4);
            memcpy(y.v, y_le, Q::N * 4);
            x = fe_to_mont<Q>(x);
            y = fe_to_mont<Q>(y);
            memcpy(xy, x.v, Q::N * 4);
            memcpy(xy + Q::N / 2, y.v, Q::N * 4);
        }
        vt->append_commitment(vt->user, label, xy, inf ? 1 : 0);
    }
    void challenge_scalar(const char* label, size_t, uint8_t out_le[32]) override {
        uint64_t m[4];
        vt->challenge_scalar(vt->user, label, m);
        HostF<R>::to_le_bytes(HostF<R>::from_words(m), out_le);
    }
};
}  // namespace zkt

// The two grand products alone (rows a8 / a9 of SURVEY.md section 8: permutation/mod.rs:181-254, lookup/mod.rs:94-151) over
// the loaded circuit's sigma evaluations and domain: the same launches round 3 of the prover makes (term kernels, prefix
// and suffix product scans, ONE host inversion for both denominators, z_combine), on caller-supplied vectors.
template <class C>
static int debug_grand_products_t(zkt_ctx* c, const uint64_t* ch, const uint64_t* const* v, uint64_t* out_z1, uint64_t* out_z2) {
    using R = typename C::Fr;
    using F = Fe<R>;
    CircuitState& S = *c->circuit;
    const size_t n = S.n;
    void* dst[7] = {S.ev[0], S.ev[1], S.ev[2], S.ev[4], S.ev[3], S.ev[5], S.ev[6]};   // a b c f t h1 h2
    for (int k = 0; k < 7; ++k) ZKT_HIP(c, hipMemcpyAsync(dst[k], v[k], n * 32, hipMemcpyHostToDevice, c->stream));
    ZTermsArgs za{};
    za.a = S.ev[0]; za.b = S.ev[1]; za.c = S.ev[2];
    za.s1 = S.sigma_ev[0]; za.s2 = S.sigma_ev[1]; za.s3 = S.sigma_ev[2]; za.roots = S.roots;
    za.f = S.ev[4]; za.t = S.ev[3]; za.h1 = S.ev[5]; za.h2 = S.ev[6];
    za.num = S.sc[0]; za.den = S.sc[1]; za.n = n;
    memcpy(za.beta, ch, 32); memcpy(za.gamma, ch + 4, 32); memcpy(za.delta, ch + 8, 32); memcpy(za.epsilon, ch + 12, 32);
    void* pn2 = S.wcos[W_Z1];
    void* sd2 = (char*)S.wcos[W_Z1] + n * 32;
    F* pin = (F*)S.pinned;
    int rc;
    if ((rc = z1_terms(c, za))) return rc;
    if ((rc = scan_mul(c, S.sc[0], S.sc[2], n, false, S.scan_tmp))) return rc;
    if ((rc = scan_mul(c, S.sc[1], S.sc[3], n, true, S.scan_tmp))) return rc;
    if ((rc = z2_terms(c, za))) return rc;
    if ((rc = scan_mul(c, S.sc[0], pn2, n, false, S.scan_tmp))) return rc;
    if ((rc = scan_mul(c, S.sc[1], sd2, n, true, S.scan_tmp))) return rc;
    ZKT_HIP(c, hipMemcpyAsync(pin, S.sc[3], 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(pin + 1, sd2, 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    if (fe_is_zero<R>(pin[0])) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "zero denominator in the permutation grand product");
    if (fe_is_zero<R>(pin[1])) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "zero denominator in the lookup grand product");
    const F d1 = pin[0], d2 = pin[1];
    const F inv12 = fe_inv_host<R>(fe_mul<R>(d1, d2));
    const F inv1 = fe_mul<R>(inv12, d2), inv2 = fe_mul<R>(inv12, d1);
    if ((rc = z_combine(c, S.sc[2], S.sc[3], inv1.v, S.ev[7], n))) return rc;
    if ((rc = z_combine(c, pn2, sd2, inv2.v, S.sc[0], n))) return rc;
    ZKT_HIP(c, hipMemcpyAsync(out_z1, S.ev[7], n * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(out_z2, S.sc[0], n * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

// A file the reference CLI wrote with --epk (bin/src/main.rs:108-109: ExtendedProverKey<F>, keys/mod.rs:148-174) against the
// extended key this circuit's ProverKey gives on the device (keys/mod.rs:78-146 as zkt_circuit_load runs it): every vector is
// recomputed in arkworks' own form (the resident cosets partly live in the quotient kernel's radix), turned canonical,
// brought to the host and compared with the file's bytes as they stream by.
template <class C>
static int circuit_check_epk_t(zkt_ctx* c, const char* path, int* vec_out, size_t* at_out) {
    using R = typename C::Fr;
    using F = Fe<R>;
    CircuitState& S = *c->circuit;
    const size_t n = S.n, m = 4 * n;
    const int log_n = S.log_n;
    *vec_out = -1;
    *at_out = 0;
    EpkReader rd;
    struct Closer {
        EpkReader& r;
        ~Closer() { r.close(); }
    } closer{rd};
    if (!rd.open(path)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    void* plain = S.qev;          // 4n work buffers, free between proofs
    void* canon = S.wcos[W_A];
    const F one = fe_one<R>();
    uint8_t zh4[4][32];           // zh_coset takes four values: g^n w4^(i mod 4) - 1 (keys/mod.rs:115-117)
    {
        const F g = fe_from_u32<R>(R::GENERATOR), w4n = root_of_unity<R>(log_n + 2);
        const F w4 = fe_pow_u64<R>(w4n, (uint64_t)n);
        F cur = fe_pow_u64<R>(g, (uint64_t)n);
        for (int j = 0; j < 4; ++j) {
            HostF<R>::to_le_bytes(fe_sub<R>(cur, one), zh4[j]);
            cur = fe_mul<R>(cur, w4);
        }
    }
    static const int pk_of[EPK_VECTORS] = {PK_QM, PK_QL, PK_QR, PK_QO, PK_QC, -1, PK_QLOOKUP, PK_QTABLE, -1, PK_S1, -1, PK_S2, -1, PK_S3,
                                           -1, -1, -1};
    std::vector<uint8_t> want, got;
    const size_t step = (size_t)1 << 16;
    got.resize(step * 32);
    int rc;
    for (int k = 0; k < EPK_VECTORS; ++k) {
        uint64_t len = 0;
        if (!rd.next(&len)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string("not an ExtendedProverKey file: ") + path);
        const bool evals = k == EPK_QLOOKUP || k == EPK_S1 || k == EPK_S2 || k == EPK_S3;
        const size_t expect = evals ? n : m;
        if (len != expect) {      // another circuit size: no element to point at
            *vec_out = k;
            *at_out = (size_t)-1;
            return ZKT_OK;
        }
        if (k != EPK_ZH_C) {
            const void* src = plain;
            if (pk_of[k] >= 0) {
                if ((rc = ntt_run(c, log_n + 2, 0, 1, S.pk[pk_of[k]], n, plain))) return rc;
            } else if (k == EPK_QLOOKUP) {
                src = S.q_lookup_ev;
            } else if (evals) {
                src = S.sigma_ev[k == EPK_S1 ? 0 : k == EPK_S2 ? 1 : 2];
            } else if (k == EPK_X_C) {
                src = S.coset[CS_X];
            } else {              // l_1_coset = coset_fft(ifft(1, 0, ..., 0)) (keys/mod.rs:119-120)
                F nn = fe_zero<R>();
                nn.v[0] = (uint32_t)(n & 0xffffffffu);
                nn.v[1] = (uint32_t)((uint64_t)n >> 32);
                const F ninv = fe_inv_host<R>(fe_to_mont<R>(nn));
                if ((rc = gen_powers(c, S.ev[0], n, one.v, ninv.v))) return rc;
                if ((rc = ntt_run(c, log_n + 2, 0, 1, S.ev[0], n, plain))) return rc;
            }
            LinCombArgs lc{};     // times the word 1 = R^-1 as a Montgomery operand: the canonical value
            lc.poly[0] = src;
            lc.len[0] = expect;
            lc.scalar[0][0] = 1;
            lc.nterms = 1;
            if ((rc = poly_lincomb(c, lc, canon, expect))) return rc;
            want.resize(expect * 32);
            ZKT_HIP(c, hipMemcpyAsync(want.data(), canon, expect * 32, hipMemcpyDeviceToHost, c->stream));
            ZKT_HIP(c, hipStreamSynchronize(c->stream));
        }
        for (size_t i = 0; i < expect; i += step) {
            const size_t cnt = std::min(step, expect - i);
            if (!rd.read(got.data(), cnt)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string("short read from ") + path);
            bool same = true;
            if (k != EPK_ZH_C) {
                same = memcmp(want.data() + i * 32, got.data(), cnt * 32) == 0;
            } else {
                for (size_t j = 0; j < cnt && same; ++j) same = memcmp(zh4[(i + j) & 3], got.data() + j * 32, 32) == 0;
            }
            if (!same) {
                size_t j = 0;
                while (memcmp(k != EPK_ZH_C ? want.data() + (i + j) * 32 : zh4[(i + j) & 3], got.data() + j * 32, 32) == 0) ++j;
                *vec_out = k;
                *at_out = i + j;
                return ZKT_OK;
            }
        }
    }
    if (!rd.at_end()) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string("bytes after the seventeenth vector of ") + path);
    return ZKT_OK;
}

extern "C" {

zkt_transcript* zkt_transcript_new(int kind, const char* label) {
    zkt_transcript* t = new zkt_transcript();
    if (kind == ZKT_TRANSCRIPT_MERLIN) {
        auto* m = new MerlinHostTranscript(label ? label : "");
        t->merlin = &m->t;
        t->impl.reset(m);
    } else if (kind == ZKT_TRANSCRIPT_ETHEREUM) {
        t->impl.reset(new EthereumHostTranscript());
    } else {
        delete t;
        return nullptr;
    }
    return t;
}
void zkt_transcript_free(zkt_transcript* t) { delete t; }
void zkt_transcript_append_u64(zkt_transcript* t, const char* label, uint64_t v) { t->impl->append_u64(label, v); }
void zkt_transcript_append_scalars(zkt_transcript* t, const char* label, const uint8_t* le32, size_t count, int single) {
    t->impl->append_scalars(label, le32, count, 32, single != 0);
}
void zkt_transcript_append_commitment(zkt_transcript* t, const char* label, const uint8_t* x_le, const uint8_t* y_le,
                                      size_t fq_bytes, int is_infinity) {
    t->impl->append_commitment(label, x_le, y_le, fq_bytes, is_infinity != 0);
}
void zkt_transcript_seed(zkt_transcript* t, uint64_t circuit_size, const uint8_t* xy_le, const uint8_t* is_infinity,
                         size_t fq_bytes) {
    static const char* labels[10] = {"q_m_commit", "q_l_commit", "q_r_commit", "q_o_commit", "q_c_commit",
                                     "sigma1_commit", "sigma2_commit", "sigma3_commit", "q_lookup_commit", "q_table_commit"};
    t->impl->append_u64("circuit_size", circuit_size);
    for (int k = 0; k < 10; ++k) {
        const uint8_t* x = xy_le + (size_t)k * 2 * fq_bytes;
        t->impl->append_commitment(labels[k], x, x + fq_bytes, fq_bytes, is_infinity && is_infinity[k]);
    }
}
void zkt_transcript_challenge_scalar(zkt_transcript* t, const char* label, int fr_bits, uint8_t out_le32[32]) {
    t->impl->challenge_scalar(label, (size_t)fr_bits, out_le32);
}
int zkt_transcript_challenge_bytes(zkt_transcript* t, const char* label, uint8_t* out, size_t len) {
    if (!t->merlin) return ZKT_ERR_INVALID_ARGUMENT;
    t->merlin->challenge_bytes(label, out, len);
    return ZKT_OK;
}
int zkt_transcript_append_message(zkt_transcript* t, const char* label, const uint8_t* msg, size_t len) {
    if (!t->merlin) return ZKT_ERR_INVALID_ARGUMENT;
    t->merlin->append_message(label, msg, len);
    return ZKT_OK;
}

int zkt_circuit_load(zkt_ctx* c, int log_n, const uint64_t* const* pk_polys, const size_t* pk_lens) {
    if (!c || !pk_polys || !pk_lens) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    if (c->curve == ZKT_CURVE_BN254) return circuit_load_t<Bn254Curve>(c, log_n, pk_polys, pk_lens);
    return circuit_load_t<Bls381Curve>(c, log_n, pk_polys, pk_lens);
}

// With a communicator attached every rank must hold exactly its zkt_shard_range of the key: a rank that loaded the
// whole key (or somebody else's slice) would make the combined commitments a multiple of the right ones.
static int check_sharded_key(zkt_ctx* c) {
    if (!c->sharded()) return ZKT_OK;
    size_t off = 0, cnt = 0, total = 0, lo = 0, hi = 0;
    zkt::msm_slice(c, &off, &cnt, &total);
    (void)zkt_shard_range(total, c->comm.vt.rank, c->comm.vt.world, &lo, &hi);
    if (off != lo || off + cnt != hi)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT,
                       "sharded proof: this rank must load powers zkt_shard_range(total, rank, world) with zkt_srs_load_slice");
    if (c->circuit && c->circuit->G != c->comm.vt.world)
        return set_err(c, ZKT_ERR_NOT_LOADED, "the circuit was loaded for another communicator");
    return ZKT_OK;
}

static int prove_impl(zkt_ctx* c, const zkt_prove_inputs* in, HostTranscript& tr, uint8_t* out, size_t cap, size_t* len) {
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (int rc0 = check_sharded_key(c)) return rc0;
    (void)hipSetDevice(c->device);
    // the Lagrange-basis table of this domain (lagrange.hip): built on the first proof after a key or circuit change
    if (!c->lagrange_off)
        if (int rc1 = lagrange_ensure(c, c->circuit->log_n)) return rc1;
    std::vector<uint8_t> proof;
    int rc;
    if (c->curve == ZKT_CURVE_BN254) {
        Prover<Bn254Curve> p(c, *c->circuit, tr);
        rc = p.run(*in, proof);
        if (rc) {   // an early return may leave staged copies in flight
            (void)hipStreamSynchronize(c->circuit->copy_stream);
            if (c->circuit->comm_stream) (void)hipStreamSynchronize(c->circuit->comm_stream);
            if (c->circuit->aux_stream) (void)hipStreamSynchronize(c->circuit->aux_stream);
            (void)hipStreamSynchronize(c->stream);
        }
    } else {
        Prover<Bls381Curve> p(c, *c->circuit, tr);
        rc = p.run(*in, proof);
        if (rc) {
            (void)hipStreamSynchronize(c->circuit->copy_stream);
            if (c->circuit->comm_stream) (void)hipStreamSynchronize(c->circuit->comm_stream);
            if (c->circuit->aux_stream) (void)hipStreamSynchronize(c->circuit->aux_stream);
            (void)hipStreamSynchronize(c->stream);
        }
    }
    if (rc) {
        // a failed proof withdraws the announcement of its successor: the next call must not issue early work from
        // pointers the caller may have released meanwhile, nor pick up half-issued early rounds
        c->circuit->has_next = false;
        c->circuit->prefetch_stage = 0;
        return rc;
    }
    if (len) *len = proof.size();
    if (proof.size() > cap) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "proof buffer too small");
    memcpy(out, proof.data(), proof.size());
    return ZKT_OK;
}

// ---- commitments of evaluation vectors (lagrange.hip) --------------------------------------------------
int zkt_ctx_set_lagrange(zkt_ctx* c, int on) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    c->lagrange_off = !on;
    ++c->msm_epoch;   // early work of an announced proof was issued under the other setting
    return ZKT_OK;
}

int zkt_lagrange_info(zkt_ctx* c, int* log_n, size_t* bases) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    const bool ready = c->circuit && lagrange_ready(c, c->circuit->log_n) && !c->lagrange_off;
    if (log_n) *log_n = ready ? c->circuit->log_n : -1;
    if (bases) *bases = ready ? lagrange_bases(c) : 0;
    return ZKT_OK;
}

int zkt_commit_evals_dev(zkt_ctx* c, const void* d_evals, const uint64_t* blinders, int k, int path, uint64_t* out_xy,
                         int* out_is_infinity) {
    if (!c || !d_evals || !out_xy || (k > 0 && !blinders)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (k < 0 || k > 3) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "0 .. 3 blinders");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (the domain comes from it)");
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    if (c->sharded()) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "not available on a sharded key");
    (void)hipSetDevice(c->device);
    CircuitState& S = *c->circuit;
    const size_t n = S.n;
    int rc;
    if (path == 1) {
        if ((rc = lagrange_ensure(c, S.log_n))) return rc;
        if (!lagrange_ready(c, S.log_n)) return set_err(c, ZKT_ERR_NOT_LOADED, "the key is too short for the Lagrange-basis table");
    } else if (path != 0) {
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "path: 0 coefficients, 1 Lagrange basis");
    }
    c->circuit->prefetch_stage = 0;   // the work buffer below belongs to round 5 of an announced proof as well
    uint32_t* d_len = S.status + 8 + 12;
    void* d_bl = (char*)S.lag_scalars + (n + 8) * 32;
    void* poly = S.poly[12];
    ZKT_HIP(c, hipMemsetAsync(d_len, 0, 4, c->stream));
    if (k) ZKT_HIP(c, hipMemcpyAsync(d_bl, blinders, (size_t)k * 32, hipMemcpyHostToDevice, c->stream));
    // poly_from_evals + add_blinders_to_poly (util.rs:63-86, prove.rs:472-483)
    if ((rc = ntt_run(c, S.log_n, 1, 0, d_evals, n, poly))) return rc;
    if ((rc = poly_trim_len(c, poly, n, d_len, (char*)poly + n * 32, 8))) return rc;
    if (k && (rc = poly_add_blinders(c, poly, d_len, d_bl, k, n + 8))) return rc;
    if (path == 0) return msm_g1_dev(c, poly, n + (size_t)k, 0, 1, out_xy, out_is_infinity);
    if ((rc = lagrange_scalars(c, d_evals, n, d_len, d_bl, k, S.roots, S.lag_scalars))) return rc;
    if ((rc = msm_begin(c, S.lag_scalars, n + (size_t)k, 0, 1, 0, 1))) return rc;
    uint64_t xy[12] = {};
    if ((rc = msm_end(c, 0, xy))) return rc;
    const size_t words = c->curve == ZKT_CURVE_BN254 ? 8 : 12;
    memcpy(out_xy, xy, words * 8);
    if (out_is_infinity) {
        bool any = false;
        for (size_t i = 0; i < words; ++i) any = any || xy[i] != 0;
        *out_is_infinity = any ? 0 : 1;
    }
    return ZKT_OK;
}

int zkt_circuit_setup(zkt_ctx* c, int log_n, const uint64_t* const* evals, const size_t* eval_lens, int evals_on_device,
                      uint64_t* out_commitments, int* out_is_infinity) {
    if (!c || !evals || !eval_lens || !out_commitments) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (!c->msm) return set_err(c, ZKT_ERR_NOT_LOADED, "no SRS loaded (zkt_srs_load)");
    (void)hipSetDevice(c->device);
    if (log_n < 3 || log_n > 26) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "circuit bound out of range");
    if (int rc0 = check_sharded_key(c)) return rc0;
    if (c->curve == ZKT_CURVE_BN254)
        return circuit_setup_t<Bn254Curve>(c, log_n, evals, eval_lens, evals_on_device, out_commitments, out_is_infinity);
    return circuit_setup_t<Bls381Curve>(c, log_n, evals, eval_lens, evals_on_device, out_commitments, out_is_infinity);
}

// Test hook: the fused quotient pass on its own (quotient_poly.rs:98-224) over the loaded circuit's key cosets and
// caller-supplied witness cosets, so that the kernel is compared point by point on inputs that satisfy nothing.
extern "C++" template <class C>
static void debug_pi_rows(const uint64_t* pi_pos, const uint64_t* pi_vals, size_t n_pi, uint32_t* tab) {
    using R = typename C::Fr;
    for (size_t i = 0; i < n_pi; ++i) {
        tab[10 * i] = (uint32_t)(4 * pi_pos[i]);
        const Fx<R> v = fx_unpack<R>(HostF<R>::from_words(pi_vals + 4 * i));
        for (int w = 0; w < 9; ++w) tab[10 * i + 1 + w] = v.l[w];
    }
}

int zkt_debug_quotient(zkt_ctx* c, const uint64_t* challenges, const uint64_t* const* wit, const uint64_t* pi_pos,
                       const uint64_t* pi_vals, size_t n_pi, uint64_t* out) {
    if (!c || !challenges || !wit || !out || (n_pi && (!pi_pos || !pi_vals)))
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    CircuitState& S = *c->circuit;
    if (S.G != 1) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "zkt_debug_quotient: whole-coset circuits only");
    if (n_pi > (size_t)QUOTIENT_PI_DIRECT_MAX) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "too many direct public inputs");
    for (size_t i = 0; i < n_pi; ++i)
        if (pi_pos[i] >= S.n) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "public input position out of range");
    (void)hipSetDevice(c->device);
    const size_t bytes = 4 * S.n * 32;
    void* d[W_COUNT] = {};
    void* d_out = nullptr;
    uint32_t* d_tab = nullptr;
    int rc = ZKT_OK;
    for (int k = 0; k < W_COUNT && !rc; ++k) {
        if (k == W_PI && n_pi) continue;
        if (!wit[k]) rc = set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null witness coset");
        else if (!(rc = dev_alloc(c, &d[k], bytes)) && hipMemcpy(d[k], wit[k], bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = set_err(c, ZKT_ERR_HIP, "copy of a witness coset failed");
    }
    if (!rc) rc = dev_alloc(c, &d_out, bytes);
    if (!rc && n_pi) {
        std::vector<uint32_t> tab(10 * n_pi);
        if (c->curve == ZKT_CURVE_BN254) debug_pi_rows<Bn254Curve>(pi_pos, pi_vals, n_pi, tab.data());
        else debug_pi_rows<Bls381Curve>(pi_pos, pi_vals, n_pi, tab.data());
        if (!(rc = dev_alloc(c, (void**)&d_tab, tab.size() * 4)) &&
            hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = set_err(c, ZKT_ERR_HIP, "copy of the public-input rows failed");
    }
    if (!rc) {
        QuotientArgs q{};
        q.a = d[W_A]; q.b = d[W_B]; q.c = d[W_C]; q.pi = d[W_PI];
        q.z1 = d[W_Z1]; q.z2 = d[W_Z2]; q.t = d[W_T]; q.h1 = d[W_H1]; q.h2 = d[W_H2];
        q.q_m = S.coset[CS_QM]; q.q_l = S.coset[CS_QL]; q.q_r = S.coset[CS_QR]; q.q_o = S.coset[CS_QO];
        q.q_c = S.coset[CS_QC]; q.q_lookup = S.coset[CS_QLOOKUP]; q.q_table = S.coset[CS_QTABLE];
        q.sigma1 = S.coset[CS_S1]; q.sigma2 = S.coset[CS_S2]; q.sigma3 = S.coset[CS_S3];
        q.x = S.coset[CS_X]; q.l1 = S.coset[CS_L1];
        q.out = d_out;
        memcpy(q.alpha, challenges, 32); memcpy(q.beta, challenges + 4, 32); memcpy(q.gamma, challenges + 8, 32);
        memcpy(q.delta, challenges + 12, 32); memcpy(q.epsilon, challenges + 16, 32);
        memcpy(q.zh_inv, S.zh_inv, sizeof(q.zh_inv));
        q.n4 = 4 * S.n;
        q.pi_tab = n_pi ? d_tab : nullptr;
        q.n_pi_direct = (uint32_t)n_pi;
        rc = quotient_pointwise(c, q);
        if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = set_err(c, ZKT_ERR_HIP, "quotient kernel failed");
        if (!rc && hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = set_err(c, ZKT_ERR_HIP, "copy back failed");
    }
    for (int k = 0; k < W_COUNT; ++k) dev_free(c, d[k]);
    dev_free(c, d_out);
    dev_free(c, d_tab);
    return rc;
}

int zkt_debug_grand_products(zkt_ctx* c, const uint64_t* challenges, const uint64_t* const* vectors, uint64_t* out_z1,
                             uint64_t* out_z2) {
    if (!c || !challenges || !vectors || !out_z1 || !out_z2) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    for (int k = 0; k < 7; ++k)
        if (!vectors[k]) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null vector");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    (void)hipSetDevice(c->device);
    c->circuit->prefetch_stage = 0;   // the work buffers are shared with an announced proof's early rounds
    if (c->curve == ZKT_CURVE_BN254) return debug_grand_products_t<Bn254Curve>(c, challenges, vectors, out_z1, out_z2);
    return debug_grand_products_t<Bls381Curve>(c, challenges, vectors, out_z1, out_z2);
}

int zkt_circuit_check_epk_file(zkt_ctx* c, const char* epk_path, int* first_mismatch_vector, size_t* mismatch_at) {
    if (!c || !epk_path || !first_mismatch_vector || !mismatch_at) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    if (c->circuit->G != 1) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "a sharded context holds one class of every coset: check the file on an unsharded one");
    (void)hipSetDevice(c->device);
    c->circuit->prefetch_stage = 0;   // the work buffers are shared with an announced proof's early rounds
    if (c->curve == ZKT_CURVE_BN254) return circuit_check_epk_t<Bn254Curve>(c, epk_path, first_mismatch_vector, mismatch_at);
    return circuit_check_epk_t<Bls381Curve>(c, epk_path, first_mismatch_vector, mismatch_at);
}

// Row a13's two kernels alone (linearization_poly.rs:55-121): k <= 12 polynomials of the prover's lengths, each evaluated at its
// own point (poly_eval_many), and their linear combination sum_j scalars[j] polys[j] (poly_lincomb)
int zkt_debug_eval_lincomb(zkt_ctx* c, const uint64_t* const* polys, const size_t* lens, int k, const uint64_t* points,
                           const uint64_t* scalars, uint64_t* out_evals, uint64_t* out_lincomb, size_t out_len) {
    if (!c || !polys || !lens || !points || !scalars || !out_evals || !out_lincomb) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    CircuitState& S = *c->circuit;
    const size_t cap = S.n + 8;
    if (k < 1 || k > 12 || out_len < 1 || out_len > cap) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "1 .. 12 polynomials, out_len <= n + 8");
    for (int j = 0; j < k; ++j)
        if (!polys[j] || lens[j] < 1 || lens[j] > cap) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "1 .. n + 8 coefficients each");
    (void)hipSetDevice(c->device);
    S.prefetch_stage = 0;   // the work buffers are shared with an announced proof's early rounds
    EvalArgs ea{};
    LinCombArgs lc{};
    ea.count = k;
    lc.nterms = k;
    for (int j = 0; j < k; ++j) {
        ZKT_HIP(c, hipMemcpyAsync(S.poly[j], polys[j], lens[j] * 32, hipMemcpyHostToDevice, c->stream));
        ea.poly[j] = lc.poly[j] = S.poly[j];
        ea.len[j] = lc.len[j] = lens[j];
        memcpy(ea.point[j], points + 4 * j, 32);
        memcpy(lc.scalar[j], scalars + 4 * j, 32);
    }
    void* d_partials = (char*)S.small + 64 * 32;
    void* d_results = (char*)S.small + 32 * 32;
    int rc = poly_eval_many(c, ea, d_partials, d_results, S.eval_pw);
    if (rc) return rc;
    if ((rc = poly_lincomb(c, lc, S.sc[0], out_len))) return rc;
    ZKT_HIP(c, hipMemcpyAsync(out_evals, d_results, (size_t)k * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(out_lincomb, S.sc[0], out_len * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

// kzg10::compute_witness_polynomial alone (row a12: the division of prove.rs:381-451's aggregated polynomial by X - z)
int zkt_debug_open_witness(zkt_ctx* c, const uint64_t* coeffs, size_t len, const uint64_t* z4, uint64_t* out) {
    if (!c || !coeffs || !z4 || !out) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    CircuitState& S = *c->circuit;
    if (len < 2 || len > S.n + 8) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "2 .. n + 8 coefficients");
    (void)hipSetDevice(c->device);
    c->circuit->prefetch_stage = 0;
    uint32_t z[8], zi[8];
    memcpy(z, z4, 32);
    bool zero = true;
    for (int i = 0; i < 8; ++i) zero = zero && z[i] == 0;
    if (zero) return set_err(c, ZKT_ERR_ZERO_DENOMINATOR, "evaluation point is zero");
    if (c->curve == ZKT_CURVE_BN254) {
        Fe<Bn254Fr> x;
        memcpy(x.v, z, 32);
        x = fe_inv_host<Bn254Fr>(x);
        memcpy(zi, x.v, 32);
    } else {
        Fe<Bls381Fr> x;
        memcpy(x.v, z, 32);
        x = fe_inv_host<Bls381Fr>(x);
        memcpy(zi, x.v, 32);
    }
    ZKT_HIP(c, hipMemcpyAsync(S.sc[0], coeffs, len * 32, hipMemcpyHostToDevice, c->stream));
    int rc = open_witness(c, S.sc[0], len, z, zi, S.sc[1], S.sc[2], S.scan_tmp, S.sc[3], S.eval_pw);
    if (rc) return rc;
    ZKT_HIP(c, hipMemcpyAsync(out, S.sc[3], (len - 1) * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

int zkt_prove_set_next(zkt_ctx* c, const zkt_prove_inputs* next) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    if (!c->circuit) return set_err(c, ZKT_ERR_NOT_LOADED, "no circuit loaded (zkt_circuit_load)");
    c->circuit->has_next = next != nullptr;
    if (next) c->circuit->next_in = *next;
    return ZKT_OK;
}

int zkt_prove(zkt_ctx* c, const zkt_prove_inputs* in, zkt_transcript* tr, uint8_t* proof_out, size_t proof_cap,
              size_t* proof_len) {
    if (!c || !in || !tr || !proof_out) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    return prove_impl(c, in, *tr->impl, proof_out, proof_cap, proof_len);
}

int zkt_prove_with(zkt_ctx* c, const zkt_prove_inputs* in, const zkt_transcript_vtable* vt, uint8_t* proof_out,
                   size_t proof_cap, size_t* proof_len) {
    if (!c || !in || !vt || !proof_out) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (c->curve == ZKT_CURVE_BN254) {
        VtableAdapter<Bn254Curve> a(vt);
        return prove_impl(c, in, a, proof_out, proof_cap, proof_len);
    }
    VtableAdapter<Bls381Curve> a(vt);
    return prove_impl(c, in, a, proof_out, proof_cap, proof_len);
}

}  // extern "C"
