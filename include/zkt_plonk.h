/*
 * zkt_plonk.h -- C-ABI of the MI355X (gfx950) prover hot path for ZKTLabs/zkt-plonk.
 *
 * The reference is pure Rust and has no FFI; the two generic seams of
 * ZKTPlonk<F, D, PC, T, C, TABLE_SIZE> (plonk-core/src/plonk.rs:39-52) are where a replacement
 * plugs in.  Each entry point below names the reference interface it replaces (file:line relative to
 * /root/reference).  INTEGRATION.md shows the Rust shim (GpuDomain<F>, GpuKZG10<E>, prove_gpu)
 * that binds them.
 *
 * Conventions:
 *  - Field elements cross as little-endian u64 limbs in MONTGOMERY form, exactly arkworks'
 *    in-memory representation (ark-ff 0.3 Fp256 / Fp384): Fr = 4 limbs for both curves,
 *    Fq = 4 limbs (BN254) or 6 limbs (BLS12-381).
 *  - A G1 affine point crosses as x limbs || y limbs; the all-zero pair (0, 0) encodes the point at
 *    infinity (GroupAffine is repr(Rust): the shim repacks explicitly, it never transmutes).
 *  - The caller owns every host buffer; pointers are borrowed for the duration of the call.
 *    Entry points with the _dev suffix take DEVICE pointers (HBM resident, same layout) and enqueue
 *    on the context's stream without synchronising.
 *  - Every call returns ZKT_OK or an error code; zkt_last_error() gives the message.  Nothing
 *    aborts: where the reference panics (zero denominators, equal challenges, short quotient) the
 *    library reports an error.
 *  - A context is used by one thread at a time; several contexts may coexist.
 */
#ifndef ZKT_PLONK_H
#define ZKT_PLONK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkt_ctx zkt_ctx;

enum {
    ZKT_CURVE_BN254 = 0,     /* ark-bn254 0.3    (bin/src/instance.rs:7-10)  */
    ZKT_CURVE_BLS12_381 = 1  /* ark-bls12-381 0.3 (bin/src/instance.rs:12-15) */
};

enum {
    ZKT_OK = 0,
    ZKT_ERR_INVALID_ARGUMENT = 1,
    ZKT_ERR_INVALID_DOMAIN_SIZE = 2,   /* Error::InvalidEvalDomainSize, plonk-core/src/error.rs */
    ZKT_ERR_HIP = 3,
    ZKT_ERR_NO_DEVICE = 4,
    ZKT_ERR_TOO_MANY_COEFFICIENTS = 5, /* kzg10 Error::TooManyCoefficients -> Error::PCError     */
    ZKT_ERR_ZERO_DENOMINATOR = 6,      /* reference: .inverse().unwrap() panics                   */
    ZKT_ERR_EQUAL_CHALLENGES = 7,      /* reference: assert_ne! at prove.rs:202-207               */
    ZKT_ERR_NOT_IN_TABLE = 8,          /* Error::ElementNotIndexedInTable, multiset.rs:121        */
    ZKT_ERR_QUOTIENT_TOO_SHORT = 9,    /* reference: slice panic at prove.rs:287-300              */
    ZKT_ERR_NOT_LOADED = 10,
    ZKT_ERR_COMM = 11                  /* the caller's communicator reported a failure */
};

/* ---- context ------------------------------------------------------------------------------ */
/* Creates a context on HIP device `device_id` for `curve_id`.  Fails with ZKT_ERR_NO_DEVICE when no
 * GPU is present: there is no CPU fallback. */
int zkt_ctx_create(int curve_id, int device_id, zkt_ctx** out);
void zkt_ctx_destroy(zkt_ctx* ctx);
/* A second context on the same GPU that SHARES `ctx`'s read-only tables -- the SRS window table and the Lagrange-basis
 * table, the circuit's keys (ProverKey polynomials, ExtendedProverKey cosets), the transform twiddles -- and owns only
 * what a proof writes: its stream, work buffers and MSM slots (about 3 of the ~7 GiB a context holds at n = 2^20).  For a
 * service that keeps several proofs in flight on one GPU (one host thread per context), which is what hides the latency
 * chain of the reference's real circuit sizes: n = 2^14 385 -> 587 proofs/s with two contexts, n = 2^18 156 -> 186 with
 * three.  The fork proves exactly as `ctx` would (same bytes).  Rules: no communicator on either side; while forks are
 * alive `ctx` refuses zkt_srs_* / zkt_circuit_* / zkt_ctx_set_comm (its tables are in use), and zkt_ctx_destroy(ctx) takes
 * effect when the last fork is destroyed; a fork that loads a key or circuit of its own simply stops sharing that part.
 * Create and destroy contexts from one thread (or serialise those calls); prove on them concurrently. */
int zkt_ctx_fork(zkt_ctx* ctx, zkt_ctx** out);
const char* zkt_last_error(const zkt_ctx* ctx);
/* Use an existing hipStream_t (e.g. PyTorch's current stream) for every launch of this context. */
int zkt_ctx_set_stream(zkt_ctx* ctx, void* hip_stream);
int zkt_ctx_synchronize(zkt_ctx* ctx);
/* Timing with HIP events on the stream the kernels run on (each event pair costs a few microseconds of stream time,
 * so only these scopes exist).  Names: "ntt_<log2 size>" (whole transform), "msm_main" (grouping, accumulation and
 * bucket fold of one MSM), "msm_accumulate" (the accumulation kernel alone), "msm_fold" / "msm_tail" (bucket fold and
 * reduction, on the side stream), "quotient", and the prover's rounds as stream time between their first and last
 * launch: "round1", "round2" (prove.rs:116-185; issued early when announced by zkt_prove_set_next), "round3"
 * (:190-255), "round4" (:258-313), "round5" (:318-451); "msm_lag_main" / "msm_lag_accumulate": the same two scopes for
 * commitments taken in the Lagrange basis (their pairs are few, they would dilute the dense kernel's average);
 * "host_wait": idle time of the context's stream across the prover's host round trips (from the moment the stream
 * drains while the host waits for a round's commitments or evaluations to the next launch; six per proof).
 * A scope that covers a batch counts its units in `calls` (the three commitments of a round grouped and accumulated as one
 * batch of launches: 3); "<name>#launches" returns the number of recorded scopes instead.
 * on = 0: off; 1: every scope; 2: only "msm_accumulate" and "host_wait" -- the level for timing the dominant kernel
 * and the stream's idle time live inside a throughput measurement (~18 event pairs per proof instead of ~80). */
int zkt_profile_enable(zkt_ctx* ctx, int on);
int zkt_profile_get(zkt_ctx* ctx, const char* name, uint64_t* calls, double* total_ms);
const char* zkt_version(void);

/* ---- one proof across the GPUs of a node (BASELINE.json configs[4]) -------------------------
 * One process (and one context) per GPU; the same calls are made on every rank with the same inputs.  What shards:
 *   - every KZG commitment (commitment.rs:24-46): rank r keeps the SRS slice zkt_shard_range(total, r, world) and sums
 *     over it; the partial sums of a prover round travel in ONE all-gather as raw bytes (world x k x 128|192 B) and are
 *     added on every rank (a collective cannot reduce curve points);
 *   - the 4n-coset work of round 4 (quotient_poly.rs:52-224): rank r owns the coset points whose index is r modulo
 *     world -- a coset of 4n / world points on which it transforms the nine witness polynomials (keys: once, at load)
 *     and runs the fused quotient pass with no communication ("omega-next" = index + 4 stays in the class for world
 *     <= 4; with 8 ranks the four shifted vectors are transformed on the neighbouring class as well).  ONE all-gather
 *     of 4n x 32 B in total brings the quotient evaluations together before the inverse transform.
 * Everything else (the n-point inverse transforms, grand products, evaluations, openings' polynomials, Fiat-Shamir) is
 * replicated: every rank produces the same proof bytes, equal to the single-GPU bytes.
 * The communicator is supplied by the caller: libzkt_comm_rccl.so (zkt_comm_rccl.h) is that communicator over RCCL as a C
 * library of its own, for hosts that have none (a Rust binary); zkt-plonk_amd/parallel.py has the same over
 * torch.distributed.  This library links no transport.  With more than one GPU the device exchange -- an in-place
 * all-gather on the library's own HBM buffer -- is UNVERIFIED ON HARDWARE: no multi-GPU node was available; it runs with
 * world-of-one RCCL communicators under 2 / 4 thread-ranks on one GPU, and the sharded prover itself is byte-checked with
 * thread and gloo ranks on one GPU up to BLS12-381 n = 2^22 x 8 ranks.
 * all_gather: `bytes` per rank, results in rank order; on_device = 0: host pointers;
 * on_device = 1 (only if device_buffers != 0): device pointers, the library has synchronised `hip_stream` before the
 * call and the exchange must be complete when the callback returns.  Returns 0 on success.
 * all_gather_async (optional, may be NULL; needs device_buffers != 0): device pointers; the collective is ENQUEUED on
 * `hip_stream` behind the work already there and the callback returns without waiting for it -- the library never
 * synchronises the host around it.  With it the quotient exchange of round 4 goes out in ZKT_QUOTIENT_CHUNKS pieces on a
 * stream of its own, each behind the kernel that produced the piece, so that the collective of one piece travels while
 * the next is computed, and the host is not blocked at any exchange of device data.  Without it the library falls back
 * to the blocking callback (one whole exchange after the quotient pass). */
typedef struct {
    void* user;
    int rank;
    int world;               /* 1, 2, 4 or 8 */
    int device_buffers;      /* 0: device exchanges are staged through pinned host memory by the library */
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes, int on_device, void* hip_stream);
    int (*all_gather_async)(void* user, const void* d_send, void* d_recv, size_t bytes, void* hip_stream);
} zkt_comm_vtable;
#define ZKT_QUOTIENT_CHUNKS 4
/* Attach (or, with NULL / world = 1, detach) the communicator.  Must precede zkt_srs_load_slice / zkt_circuit_load /
 * zkt_circuit_setup of the sharded proof: keys are laid out for the rank's share. */
int zkt_ctx_set_comm(zkt_ctx* ctx, const zkt_comm_vtable* comm);
/* Contiguous share [*lo, *hi) of `total` units for `rank` (sizes differ by at most one). */
int zkt_shard_range(size_t total, int rank, int world, size_t* lo, size_t* hi);
/* Exchanged bytes and collective calls since the communicator was attached (this rank's send side). */
int zkt_comm_stats(zkt_ctx* ctx, uint64_t* calls, uint64_t* bytes_sent);
/* Plumbing check without a GPU: gathers `bytes` from every rank through the vtable (host buffers). */
int zkt_comm_selftest(const zkt_comm_vtable* comm, const void* send, void* recv, size_t bytes);

/* ---- device memory helpers (plumbing for callers without their own allocator) -------------- */
int zkt_dev_alloc(zkt_ctx* ctx, size_t bytes, void** dptr);
int zkt_dev_free(zkt_ctx* ctx, void* dptr);
int zkt_dev_upload(zkt_ctx* ctx, void* dptr, const void* host, size_t bytes);
int zkt_dev_download(zkt_ctx* ctx, void* host, const void* dptr, size_t bytes);

/* ---- Domain seam: D: EvaluationDomain<F> + EvaluationDomainExt<F> (prove.rs:70) ------------ */
/* Radix-2 transform of size 2^log_n over Fr, natural order in and out.
 *   inverse = 0, coset = 0 : D::fft            (util.rs:104-113)  out[i] = sum_j in[j] w^(ij)
 *   inverse = 1, coset = 0 : D::ifft(_in_place) (util.rs:63-86)   inverse, scaled by 1/n
 *   inverse = 0, coset = 1 : D::coset_fft(_in_place) (util.rs:117-140)  in[j] *= g^j first
 *   inverse = 1, coset = 1 : D::coset_ifft_in_place (util.rs:90-100)    then out[j] *= g^-j
 * `in` holds in_len <= 2^log_n elements and is zero-padded (ark-poly resizes the coefficient vector);
 * `out` receives 2^log_n elements.  in == out is allowed.  log_n > TWO_ADICITY (28 / 32) or
 * in_len > 2^log_n -> ZKT_ERR_INVALID_DOMAIN_SIZE. */
int zkt_ntt(zkt_ctx* ctx, int log_n, int inverse, int coset, const uint64_t* in, size_t in_len, uint64_t* out);
int zkt_ntt_dev(zkt_ctx* ctx, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out);
/* One GPU's share of D::coset_fft on the domain of size 2^log_big (util.rs:117-140) sharded by output index over
 * G = 2^(log_big - log_n) GPUs: out[i] = p(g w_big^(cls + G i)), i < 2^log_n -- the evaluations whose index is cls
 * modulo G, a coset of their own, obtained by one 2^log_n-point transform with no exchange.  in_len may exceed 2^log_n
 * (the polynomial is folded modulo X^(2^log_n) - shift^(2^log_n) first).  Host pointers. */
int zkt_ntt_class(zkt_ctx* ctx, int log_n, int log_big, int cls, const uint64_t* in, size_t in_len, uint64_t* out);
/* EvaluationDomainExt::group_gen (util.rs:52-58): writes the 2^log_n-th root of unity (4 limbs). */
int zkt_domain_group_gen(zkt_ctx* ctx, int log_n, uint64_t* out4);

/* ---- Commitment seam: PC: HomomorphicCommitment<F> = KZG10<E> (commitment.rs:10-46) --------- */
/* Loads `count` G1 powers (ck.powers_of_g of SonicKZG10's CommitterKey, produced by PC::trim at
 * plonk.rs:79-85) and precomputes the window multiples used by the MSM.  One-time per key.  The
 * prover never commits to more than n + 7 coefficients (the opening witnesses), so loading n + 8 of the 4n + 1 powers the reference keeps is
 * enough.  Replaces nothing at run time: it is the device-resident form of `ck`. */
int zkt_srs_load(zkt_ctx* ctx, const uint64_t* g1_xy_mont, size_t count);
int zkt_srs_load_dev(zkt_ctx* ctx, const void* d_g1_xy_mont, size_t count);
/* Test/bench SRS with a KNOWN trapdoor: powers_of_g[i] = tau^i * G (tau: 4 canonical limbs).
 * Stands in for PC::setup (ark-poly-commit kzg10 setup), which is out of scope; insecure by design. */
int zkt_srs_generate(zkt_ctx* ctx, const uint64_t* tau_canonical4, size_t count);
int zkt_srs_download(zkt_ctx* ctx, size_t offset, size_t count, uint64_t* out_xy_mont);
/* The G2 half of that test SRS, i.e. SonicKZG10's VerifierKey::h and ::beta_h for the same trapdoor: h = the G2
 * generator of ark-bn254 / ark-bls12-381, beta_h = tau h (arkworks' Fp2 layout x.c0, x.c1, y.c0, y.c1, Montgomery
 * limbs).  Host-only; what zkt_verify takes next to the proof.  Insecure by design, like zkt_srs_generate. */
int zkt_srs_generate_g2(int curve_id, const uint64_t* tau_canonical4, uint64_t* out_h, uint64_t* out_beta_h);
/* Sharded committer key: this rank keeps powers [offset, offset + count) of a key of `total` powers -- its
 * zkt_shard_range(total, rank, world).  The window table shrinks by the number of ranks.  zkt_msm_g1* then index into the
 * slice; zkt_prove / zkt_circuit_setup combine the ranks' partial sums through the communicator. */
int zkt_srs_load_slice(zkt_ctx* ctx, const uint64_t* g1_xy_mont_slice, size_t offset, size_t count, size_t total);
int zkt_srs_generate_slice(zkt_ctx* ctx, const uint64_t* tau_canonical4, size_t offset, size_t count, size_t total);
/* sum_i scalars[i] * powers_of_g[base_offset + i], affine result (x || y Montgomery limbs, (0,0) and
 * *out_is_infinity = 1 for the identity).  This is VariableBaseMSM::multi_scalar_mul as called by
 * kzg10::commit / open_with_witness_polynomial (prove.rs:133-135,178-180,249-251,306-308,373-375,
 * 381-451): scalars_montgomery = 1 takes polynomial coefficients as they sit in a DensePolynomial
 * (the into_repr() conversion happens on the device), 0 takes canonical bigints (commitment.rs:36-42).
 * base_offset mirrors skip_leading_zeros_and_convert_to_bigints (powers_of_g[num_leading_zeros..]).
 * len + base_offset > loaded powers -> ZKT_ERR_TOO_MANY_COEFFICIENTS; len = 0 -> identity. */
int zkt_msm_g1(zkt_ctx* ctx, const uint64_t* scalars, size_t len, size_t base_offset, int scalars_montgomery,
               uint64_t* out_xy_mont, int* out_is_infinity);
/* Same with the scalars already resident in HBM; the affine result is written to HOST memory
 * (it feeds the host-side transcript).  Synchronises the stream. */
int zkt_msm_g1_dev(zkt_ctx* ctx, const void* d_scalars, size_t len, size_t base_offset, int scalars_montgomery,
                   void* out_xy_mont_host);
/* Enqueue-only form for benchmarking the device part under HIP events (no host finish, no sync). */
int zkt_msm_enqueue_dev(zkt_ctx* ctx, const void* d_scalars, size_t len, size_t base_offset, int scalars_montgomery);
/* Window size c, number of windows and loaded powers of the current SRS (0s when none). */
int zkt_msm_info(zkt_ctx* ctx, int* window_bits, int* windows, size_t* srs_count);

/* ---- Commitments of evaluation vectors (Lagrange-basis key) ------------------------------------------
 * The reference commits to t, h1, h2 and z2 through their coefficients (prove.rs:145-180,225-251: poly_from_evals,
 * add_blinders_to_poly, PC::commit -- one dense MSM each).  As EVALUATION vectors they are piecewise constant (table
 * values then zeros, sorted runs, a grand product whose ratio is 1 wherever the lookup stands still), so the library
 * commits to them in the Lagrange basis of the circuit's domain: with S_k = sum_{i<k} [L_i(tau)] G,
 *     commit = sum_k (e_(k-1) - e_k) S_k + sum_j b_j ([tau^(n+j)] G - [tau^j] G),
 * an MSM whose scalars vanish inside every run.  The same group element, so the same proof bytes; a dense vector costs
 * what its coefficients would.  The second base table (prefix sums of the inverse DFT of the powers over G1, plus the
 * blinder points; as large as the first) is built on the first proof after a key or circuit change -- about 0.55 s at
 * n = 2^20 on BN254 -- when the key is whole (not a slice of a sharded key) and holds more than n powers; otherwise,
 * or after zkt_ctx_set_lagrange(ctx, 0), the coefficients are committed as the reference does. */
int zkt_ctx_set_lagrange(zkt_ctx* ctx, int on);
/* *log_n = domain the table serves (-1: none, evaluations go through their coefficients), *bases = its points */
int zkt_lagrange_info(zkt_ctx* ctx, int* log_n, size_t* bases);
/* PC::commit of poly_from_evals(domain, evals) (util.rs:63-86) with k in 0..3 blinders added as add_blinders_to_poly
 * does (prove.rs:472-483); the domain is the loaded circuit's.  d_evals: n elements in HBM; blinders: k x 4 words,
 * host.  path 0 = through the coefficients (the reference's route), 1 = through the Lagrange-basis table (built if
 * need be; ZKT_ERR_NOT_LOADED when the key cannot carry one).  Both give the same affine point (x || y Montgomery
 * limbs on the host, (0,0) and *out_is_infinity = 1 for the identity).  Synchronises the stream. */
int zkt_commit_evals_dev(zkt_ctx* ctx, const void* d_evals, const uint64_t* blinders, int k, int path, uint64_t* out_xy_mont,
                         int* out_is_infinity);

/* ---- Fiat-Shamir transcripts (host side; T: TranscriptProtocol, plonk-core/src/transcript.rs:16-45) */
enum {
    ZKT_TRANSCRIPT_MERLIN = 0,   /* MerlinTranscript, plonk-core/src/transcript.rs:46-109 (merlin 3.0) */
    ZKT_TRANSCRIPT_ETHEREUM = 1  /* EthereumTranscript, gadgets/src/transcript.rs:8-90 (BN254 only)   */
};
typedef struct zkt_transcript zkt_transcript;
/* T::new(label) (plonk.rs:105).  Scalars/coordinates are canonical little-endian bytes here. */
zkt_transcript* zkt_transcript_new(int kind, const char* label);
void zkt_transcript_free(zkt_transcript* t);
void zkt_transcript_append_u64(zkt_transcript* t, const char* label, uint64_t v);            /* transcript.rs:58-60 */
/* count scalars of 32 bytes; single != 0 mirrors append_scalar, 0 mirrors append_scalars (transcript.rs:62-79) */
void zkt_transcript_append_scalars(zkt_transcript* t, const char* label, const uint8_t* le32, size_t count, int single);
void zkt_transcript_append_commitment(zkt_transcript* t, const char* label, const uint8_t* x_le, const uint8_t* y_le,
                                      size_t fq_bytes, int is_infinity);                      /* transcript.rs:81-86 */
void zkt_transcript_challenge_scalar(zkt_transcript* t, const char* label, int fr_bits, uint8_t out_le32[32]); /* :101-108 */
/* VerifierKey::seed_transcript (proof_system/keys/mod.rs:260-275) in one call: circuit_size, then the ten commitments
 * q_m q_l q_r q_o q_c sigma1 sigma2 sigma3 q_lookup q_table under their "<name>_commit" labels.  xy_le: 10 x (x, y),
 * fq_bytes little-endian canonical bytes per coordinate; is_infinity: 10 flags (may be NULL = none). */
void zkt_transcript_seed(zkt_transcript* t, uint64_t circuit_size, const uint8_t* xy_le, const uint8_t* is_infinity,
                         size_t fq_bytes);
/* raw merlin access (conformance vectors); ZKT_ERR_INVALID_ARGUMENT on a non-merlin transcript */
int zkt_transcript_append_message(zkt_transcript* t, const char* label, const uint8_t* msg, size_t len);
int zkt_transcript_challenge_bytes(zkt_transcript* t, const char* label, uint8_t* out, size_t len);

/* Foreign transcript: the Rust shim implements these four callbacks on top of its own
 * `T: TranscriptProtocol<F, PC::Commitment>`; values cross as arkworks Montgomery limbs. */
typedef struct {
    void* user;
    void (*append_u64)(void* user, const char* label, uint64_t v);
    void (*append_scalars)(void* user, const char* label, const uint64_t* fr_mont, size_t count, int single);
    void (*append_commitment)(void* user, const char* label, const uint64_t* g1_xy_mont, int is_infinity);
    void (*challenge_scalar)(void* user, const char* label, uint64_t* fr_mont_out4);
} zkt_transcript_vtable;

/* ---- Prover: proof_system::prove (plonk-core/src/proof_system/prove.rs:59-470) ---------------- */
/* Loads the preprocessed circuit: the ten ProverKey polynomials in coefficient form
 * (keys/mod.rs:29-77), order q_m, q_l, q_r, q_o, q_c, sigma1, sigma2, sigma3, q_lookup, q_table, each
 * pk_lens[k] <= n = 2^log_n coefficients (trailing zeros stripped or not).  The ExtendedProverKey
 * (keys/mod.rs:78-174: 13 coset vectors on the 4n domain, sigma / q_lookup evaluations) is derived on
 * the device and stays resident in HBM; it never crosses PCIe. */
int zkt_circuit_load(zkt_ctx* ctx, int log_n, const uint64_t* const* pk_polys, const size_t* pk_lens);

/* proof_system::setup on the device (plonk-core/src/proof_system/setup.rs:42-166): the ten padded evaluation vectors
 * of the SetupComposer in ProverKey order (q_m, q_l, q_r, q_o, q_c, sigma1, sigma2, sigma3, q_lookup, q_table;
 * eval_lens[k] <= n values each, the rest zero - setup.rs:28-35 pad_to) become the ProverKey polynomials (iNTT), the
 * ExtendedProverKey (as in zkt_circuit_load) and the ten VerifierKey commitments (PC::commit, setup.rs:104-121).
 * out_commitments: 10 x (x, y) in arkworks Montgomery limbs (2 x 4 u64 on BN254, 2 x 6 on BLS12-381), (0, 0) and
 * out_is_infinity[k] = 1 for the identity (an all-zero selector).  The circuit is left loaded: zkt_prove can follow.
 * Needs zkt_srs_load with >= n powers.  The permutation bookkeeping that produces the sigma evaluations
 * (permutation/mod.rs compute_all_sigma_evals) stays with the caller. */
int zkt_circuit_setup(zkt_ctx* ctx, int log_n, const uint64_t* const* evals, const size_t* eval_lens, int evals_on_device,
                      uint64_t* out_commitments, int* out_is_infinity);

typedef struct {
    /* wire_evals() of the proving composer (prove.rs:49-55,116): n_rows <= n values each, zero padded */
    const uint64_t* a_evals;
    const uint64_t* b_evals;
    const uint64_t* c_evals;
    size_t n_rows;
    /* the lookup table as a LookupTable IndexSet (lookup/table.rs:19): distinct values, insertion order */
    const uint64_t* table;
    size_t table_len;
    /* PublicInputs BTreeMap (constraint_system/pi.rs:52-105): ascending positions and their values */
    const size_t* pi_pos;
    const uint64_t* pi_vals;
    size_t n_pi;
    /* the 19 F::rand(rng) draws of prove.rs in reference order:
     * a(2) b(2) c(2) h1(3) h2(2) z1(3) z2(3) b0 b1   (prove.rs:125-127,170-171,225,244,296) */
    const uint64_t* blinders;
    /* non-zero: a_evals / b_evals / c_evals are DEVICE pointers (witness already resident in HBM) */
    int wires_on_device;
    /* Alternative to the three evaluation vectors, used when a_evals is NULL: the witness as the composer holds it
     * (prove.rs:49-55 wire_evals runs on the device).  `variables`: n_vars assigned values (var_map, Montgomery
     * form); w_l / w_r / w_o: n_rows variable indices per gate, ZKT_VARIABLE_ZERO = Variable::Zero (value 0,
     * constraint_system/variable.rs:10-15).  An index >= n_vars -> ZKT_ERR_INVALID_ARGUMENT.  wires_on_device
     * applies to these four arrays as well. */
    const uint64_t* variables;
    size_t n_vars;
    const uint32_t* w_l;
    const uint32_t* w_r;
    const uint32_t* w_o;
} zkt_prove_inputs;
#define ZKT_VARIABLE_ZERO 0xFFFFFFFFu

/* Runs the five prover rounds on the device and writes the CanonicalSerialize bytes of
 * Proof<F, D, KZG10<E>> (proof.rs:106-155): 802 bytes on BN254, 1010 on BLS12-381.  The transcript must
 * already be seeded by VerifierKey::seed_transcript (keys/mod.rs:260-275), as in plonk.rs:105-108.
 * Needs zkt_srs_load (>= n + 8 powers) and zkt_circuit_load.  Errors mirror the reference's Err / panics:
 * ZKT_ERR_NOT_IN_TABLE, ZKT_ERR_EQUAL_CHALLENGES, ZKT_ERR_ZERO_DENOMINATOR, ZKT_ERR_QUOTIENT_TOO_SHORT. */
int zkt_prove(zkt_ctx* ctx, const zkt_prove_inputs* in, zkt_transcript* transcript, uint8_t* proof_out,
              size_t proof_cap, size_t* proof_len);
/* Optional, for back-to-back proofs on one context: announces the inputs of the proof that will follow the next
 * zkt_prove / zkt_prove_with call.  Rounds 1 and 2 (prove.rs:116-185) need no challenge, so that call issues them for
 * `next` behind its own quotient commitments (round 1) and opening commitments (round 2): their bucket reductions
 * are latency-bound and leave the GPU nearly empty otherwise, and it no longer drains between the two proofs.  The following zkt_prove must
 * be given the same inputs (same pointers and sizes; the data they point to must stay unchanged meanwhile) - otherwise
 * the early work is simply redone.  NULL withdraws the announcement.  Proof bytes are the same either way. */
int zkt_prove_set_next(zkt_ctx* ctx, const zkt_prove_inputs* next);
int zkt_prove_with(zkt_ctx* ctx, const zkt_prove_inputs* in, const zkt_transcript_vtable* transcript,
                   uint8_t* proof_out, size_t proof_cap, size_t* proof_len);

/* ---- Witness synthesis for Poseidon-heavy circuits as batched field kernels ----------------------
 * (1) The permutation of plonk-hashing/src/hasher/poseidon/spec.rs (rounds :18-111, schedule :267-316, input layout
 * :239-265: state[0] = domain_tag, the inputs follow, output = state[1]) for `batch` independent hashes, one thread per
 * hash: what NativePlonkSpecRef computes.  out_states (optional) receives every round's state, batch x (rounds + 1) x
 * width scalars -- ROUND STATES ONLY: they are not the variables of the in-circuit gadget (that is (2) below).
 * Constants are the caller's PoseidonConstants (the reference generates them at run time, constants.rs:27, or parses the
 * BN254 tables of gadgets/src/poseidon).  Everything in Montgomery limbs; host pointers. */
typedef struct {
    int width;                       /* 2 .. 8 */
    int half_full_rounds;            /* full rounds before and after the partial ones */
    int partial_rounds;
    const uint64_t* round_constants; /* (2 half_full + partial) x width */
    const uint64_t* mds;             /* width x width, row major: m[i][j] */
    const uint64_t* domain_tag;      /* one scalar */
} zkt_poseidon_params;
int zkt_poseidon_hash_batch(zkt_ctx* ctx, const zkt_poseidon_params* params, const uint64_t* inputs, size_t batch, int arity,
                            uint64_t* out_hashes, uint64_t* out_states);
/* The same with everything resident in HBM: zkt_poseidon_load uploads the parameters once (as the kernel's own 29-bit
 * limbs), zkt_poseidon_hash_batch_dev takes DEVICE pointers, allocates nothing and enqueues on the context's stream
 * without synchronising.  d_out_states: optional, batch x (rounds + 1) x width scalars (round states, see above).
 * half_full_rounds >= 1 and partial_rounds >= 1 (output_hash, spec.rs:267-316, always runs one of each before its loops). */
typedef struct zkt_poseidon zkt_poseidon;
int zkt_poseidon_load(zkt_ctx* ctx, const zkt_poseidon_params* params, zkt_poseidon** out);
void zkt_poseidon_free(zkt_ctx* ctx, zkt_poseidon* params);
int zkt_poseidon_hash_batch_dev(zkt_ctx* ctx, const zkt_poseidon* params, const void* d_inputs, size_t batch, int arity,
                                void* d_out_hashes, void* d_out_states);

/* (2) The WITNESS of the in-circuit gadget PoseidonRef<ConstraintSystem, PlonkSpecRef, _, WIDTH>::hash (spec.rs:174-219,
 * 343-375).  While the reference's composer synthesises one hash it assigns a fresh variable per gate
 * (constraint_system/arithmetic.rs:19,79; variable.rs:117-126): power_of_5 is three mul_gates (x^2, x^4, x^5;
 * spec.rs:107-111), every term of product_mds an add_gate (the W^2 running sums, j outer, i inner; spec.rs:73-88);
 * add_constant / mul_constant allocate nothing (lazy LTVariable transforms, variable.rs:77-86).  That is
 *     zkt_poseidon_gadget_vars_per_hash = 2 half_full (3 W + W^2) + partial (3 + W^2)      (804 / 1288 / 1888 for x3 / x4 / x5)
 * variables per hash, consecutive in VariableMap::values.  This entry point computes exactly those values for `batch`
 * independent hashes and writes them, in allocation order, into the variable map the prover gathers its wires from
 * (zkt_prove_inputs.variables with wires_on_device = 1): hash h fills d_variables[base .. base + vars_per_hash) with
 * base = d_trace_base[h], or trace_base0 + h * vars_per_hash when d_trace_base is NULL.  The hash value (elements[1] after
 * the last round, spec.rs:315) is the variable at base + vars_per_hash - 1 - (W - 2) W; d_out_hashes (optional) receives
 * it as well.  Inputs are plain variables (coeff 1, offset 0: every call site of circuits/src/withdraw.rs and
 * plonk-hashing/src/merkle/binary.rs): their VALUES in d_inputs (batch x arity), or their INDICES into d_variables in
 * d_input_vars (ZKT_VARIABLE_ZERO = Variable::Zero) -- exactly one of the two.  Hashes of one launch are independent:
 * an input may not be a variable the same launch writes.  Device pointers, no allocation, no synchronisation; a trace
 * base or input index outside [0, n_vars) makes the kernel skip that hash and raise a flag that
 * zkt_poseidon_gadget_check (which synchronises the stream) turns into ZKT_ERR_INVALID_ARGUMENT.
 * "Parity unpinned": the reference holds no known answer for the gadget; tests compare with the oracle's gate-by-gate
 * restatement of the composer. */
typedef struct {
    size_t batch;
    int arity;                      /* inputs per hash, <= width - 1 */
    const void* d_inputs;           /* batch x arity scalars, or NULL */
    const uint32_t* d_input_vars;   /* batch x arity variable indices, or NULL */
    void* d_variables;              /* VariableMap::values, n_vars scalars */
    size_t n_vars;
    const uint32_t* d_trace_base;   /* per hash, or NULL */
    size_t trace_base0;
    void* d_out_hashes;             /* optional: batch scalars */
    int kernel;                     /* 0: chosen by batch size; 1: one thread per hash (throughput: large batches);
                                     * 2: width^2 lanes per hash (latency: a single proof's few hundred hashes) */
} zkt_poseidon_gadget_args;
size_t zkt_poseidon_gadget_vars_per_hash(const zkt_poseidon* params);
int zkt_poseidon_gadget_witness_dev(zkt_ctx* ctx, const zkt_poseidon* params, const zkt_poseidon_gadget_args* args);
int zkt_poseidon_gadget_check(zkt_ctx* ctx, const zkt_poseidon* params);
/* Optional validation of ONE launch's arguments on the host (downloads the index vectors, synchronises): the traces must
 * be pairwise disjoint and inside the map and no input index may be a variable the same launch writes -- the two rules
 * zkt_poseidon_gadget_witness_dev cannot afford to check per proof and whose violation yields a stale or racy witness
 * (the proof then fails to verify).  ZKT_ERR_INVALID_ARGUMENT names the rule.  Run it once per circuit layout. */
int zkt_poseidon_gadget_validate(zkt_ctx* ctx, const zkt_poseidon* params, const zkt_poseidon_gadget_args* args);

/* ---- Verifier (proof_system/proof.rs:285-503): zkt_verify_prepare = everything but the pairings,
 * ---- zkt_pairing_product_is_one = the pairings, zkt_verify = both ------------------------------------------------
 * Deserialises the proof (proof.rs:98-155; points are decompressed and checked to be on the curve), replays the
 * transcript, computes r0 (proof.rs:163-217) and the linearisation commitment (proof.rs:220-282, the 13-point
 * multi_scalar_mul of commitment.rs:32-45), and folds each of the two SonicKZG10::check calls (proof.rs:420-500) into
 * ONE pair of G1 points (L, W) with L = sum_i eta^i C_i - (sum_i eta^i v_i) g + z W: the opening is valid iff
 * e(L, h) == e(W, beta h).  The two pairings stay with the caller (arkworks), who holds h and beta h.  Host-only code:
 * a proof is ~30 short scalar multiplications.  `transcript` must be seeded like the prover's.
 * out_pairs: L1, W1, L2, W2 as (x, y) Montgomery limbs; out_is_infinity: 4 flags (may be NULL).
 * Errors: ZKT_ERR_INVALID_ARGUMENT for malformed bytes / points off the curve, ZKT_ERR_EQUAL_CHALLENGES (proof.rs:340-345). */
typedef struct {
    uint64_t n;                         /* VerifierKey::n */
    const uint64_t* vk_commitments;     /* 10 x (x, y), zkt_transcript_seed order */
    const int* vk_is_infinity;          /* 10 flags or NULL */
    const uint64_t* pi_roots;           /* VerifierKey::pi_roots, n_pi Montgomery scalars */
    const uint64_t* pub_inputs;         /* the public inputs, same order */
    size_t n_pi;
    const uint8_t* proof;               /* CanonicalSerialize bytes (802 / 1010) */
    size_t proof_len;
    const uint64_t* g;                  /* SonicKZG10 VerifierKey::g (= powers_of_g[0]), (x, y) */
} zkt_verify_inputs;
int zkt_verify_prepare(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, uint64_t* out_pairs,
                       int* out_is_infinity);

/* The pairing check itself, on the host: prod_i e(P_i, Q_i) == 1 for P_i in G1 ((x, y) Montgomery limbs) and Q_i in G2
 * (arkworks' Fp2 layout: x.c0, x.c1, y.c0, y.c1; all-zero = infinity).  The optimal ate pairing of ark-ec (Miller loop over
 * 6x + 2 plus two Frobenius steps on BN254, over |x| on BLS12-381, csrc/pairing.hpp); the line slopes of a G2 point are
 * computed once and kept, so the fixed h / beta h of a KZG check cost no G2 arithmetic per call; one shared squaring chain
 * and one final exponentiation for the whole product.  ~0.5 ms for the two pairings of one KZG check on BN254.  Points
 * off the curve / twist, G1 points outside the prime-order subgroup (BLS12-381) -> ZKT_ERR_INVALID_ARGUMENT; G2
 * subgroup membership is the caller's business (h and beta h come from the trusted VerifierKey). */
int zkt_pairing_product_is_one(int curve_id, const uint64_t* g1_xy_mont, const uint64_t* g2_xy_mont, size_t n, int* is_one);
/* The whole of Proof::verify (proof_system/proof.rs:285-503): zkt_verify_prepare, then both openings' checks
 * e(L_k, h) e(-W_k, beta h) == 1 (proof.rs:441,479) folded into one product of two pairings with a 128-bit challenge rho
 * hashed from (L1, W1, L2, W2, h, beta h): e(L1 + rho L2, h) e(-(W1 + rho W2), beta h) == 1 -- ark-poly-commit's batch_check
 * folding; a proof failing either check passes with probability 2^-128.  h, beta_h: SonicKZG10 VerifierKey::h and
 * ::beta_h (G2).  *accepted = 1 / 0 (Error::ProofVerificationError).  ~2.3 ms per BN254 proof on one host core. */
int zkt_verify(int curve_id, const zkt_verify_inputs* in, zkt_transcript* transcript, const uint64_t* h_g2_mont,
               const uint64_t* beta_h_g2_mont, int* accepted);

/* `count` proofs under ONE structured reference string (h, beta h), circuits / verifier keys free to differ: each proof is
 * taken through zkt_verify_prepare with its own seeded transcript, and all 2 * count opening checks are folded into ONE
 * product of two pairings with 128-bit coefficients hashed from every (L, W), h and beta h (see csrc/verify.hip).
 * *accepted = 1 iff every proof verifies (a batch holding a bad proof passes with probability 2^-128); a rejected batch
 * does not say which proof failed -- fall back to zkt_verify.  Cost per proof: the transcript, r0 and the two short
 * multi-scalar multiplications; the pairings (~0.5 ms on BN254) are paid once per batch.  The reference verifies proof by
 * proof (proof.rs:285-503); this is the batch form SURVEY.md 8f.4 lists.  Errors as zkt_verify_prepare (a malformed
 * proof fails the call, not just the batch). */
int zkt_verify_batch(int curve_id, const zkt_verify_inputs* ins, zkt_transcript* const* transcripts, size_t count,
                     const uint64_t* h_g2_mont, const uint64_t* beta_h_g2_mont, int* accepted);

/* HomomorphicCommitment::multi_scalar_mul (commitment.rs:32-45) for ARBITRARY points: the verifier's 13-point
 * linearisation commitment and similar short combinations.  Host arithmetic (double-and-add on 64-bit limbs): at this
 * size a device launch would cost more than the sum.  scalars: 4 limbs each, Montgomery or canonical. */
int zkt_g1_msm_host(int curve_id, const uint64_t* points_xy_mont, const uint64_t* scalars, size_t n, int scalars_montgomery,
                    uint64_t* out_xy_mont, int* out_is_infinity);

/* ---- key files of the reference CLI ------------------------------------------------------------
 * `serialize_to_file` = CanonicalSerialize::serialize_unchecked (bin/src/parser.rs:14-22).  Layouts restated from
 * ark-serialize / ark-poly-commit 0.3 (see csrc/keyfile.hip); the reference holds no key file, so these readers are
 * pinned only by the round trip against the writer in oracle/keyfile.py ("parity unpinned").  Host-only; values come
 * back as arkworks Montgomery limbs, ready for zkt_srs_load / zkt_circuit_load / zkt_transcript_seed. */
/* --ck (bin/src/main.rs:105): SonicKZG10 CommitterKey -> powers_of_g[0 .. min(count, max_powers)); max_powers = 0: all.
 * out_xy_mont = NULL: only *n_powers.  The prover needs n + 8 of the 4n + 1 powers the file holds. */
int zkt_keyfile_committer_key(const char* path, int curve_id, size_t max_powers, uint64_t* out_xy_mont, size_t* n_powers);
/* --pk (bin/src/main.rs:107): ProverKey<F> (keys/mod.rs:29-41) -> the ten coefficient vectors in zkt_circuit_load
 * order and their lengths; out_polys = NULL (or an entry NULL): lengths only. */
int zkt_keyfile_prover_key(const char* path, int curve_id, uint64_t* const* out_polys, size_t* lens10);
/* --vk (bin/src/main.rs:111): VerifierKey (keys/mod.rs:180-210) -> n, pi_roots, the ten commitments (x, y) in
 * zkt_transcript_seed order with their infinity flags. */
int zkt_keyfile_verifier_key(const char* path, int curve_id, uint64_t* n, uint64_t* pi_roots_mont, size_t pi_cap, size_t* n_pi,
                             uint64_t* commitments_xy_mont, int* is_infinity10);
/* --epk (bin/src/main.rs:34-35,108-109): ExtendedProverKey<F> (keys/mod.rs:148-174) = seventeen Vec<F> in declaration
 * order: arith { q_m_coset q_l_coset q_r_coset q_o_coset q_c_coset }, lookup { q_lookup q_lookup_coset q_table_coset },
 * perm { sigma1 sigma1_coset sigma2 sigma2_coset sigma3 sigma3_coset x_coset }, zh_coset, l_1_coset (4n values each, n
 * for q_lookup and the three sigma vectors).  This library never NEEDS the file: zkt_circuit_load derives the extended
 * key from the ProverKey on the device in milliseconds, where the file of an n = 2^20 circuit holds 1.9 GB.  It is read
 * so that a file the reference wrote can be checked against that derivation.
 * zkt_keyfile_extended_prover_key: the seventeen lengths, and vector `which` (0 .. 16; -1: lengths only) as Montgomery
 * limbs into out_mont (cap elements); streamed, nothing else is held in memory.
 * zkt_circuit_check_epk_file: every vector of the file against the loaded circuit's own extended key, recomputed on the
 * device.  *first_mismatch_vector = -1: identical; otherwise the vector (0 .. 16) and *mismatch_at the first differing
 * element, or (size_t)-1 when the vector's length is not this circuit's.  Unsharded contexts only; not during a proof. */
#define ZKT_EPK_VECTORS 17
int zkt_keyfile_extended_prover_key(const char* path, int curve_id, int which, uint64_t* out_mont, size_t cap, size_t* lens17);
int zkt_circuit_check_epk_file(zkt_ctx* ctx, const char* epk_path, int* first_mismatch_vector, size_t* mismatch_at);
/* the two loads a prover service does at start-up, straight from the CLI's files */
int zkt_srs_load_file(zkt_ctx* ctx, const char* ck_path, size_t max_powers);
int zkt_circuit_load_file(zkt_ctx* ctx, const char* pk_path, int log_n);

/* ---- debug / test support ------------------------------------------------------------------ */
/* Dumps the compiled-in parameter tables (modulus, -p^-1 mod 2^32, R, R^2) as u32 words for
 * which = 0 (Fr) or 1 (Fq) of the context's curve; returns the limb count. */
int zkt_debug_params(zkt_ctx* ctx, int which, uint32_t* out, size_t out_words);
/* Runs a field routine on the HOST (the same __host__ __device__ code the kernels execute) so that
 * the arithmetic can be pinned against big integers without a GPU.  which: 0 = Fr, 1 = Fq; a, b, out:
 * packed Montgomery words (8 or 12 x u32).  op 0: product; 1: 32-bit-limb reference product;
 * 2: arkworks form -> 29-bit limbs -> arkworks form; 3: 3*(a^2 - b^2) through the lazy add/sub/mul path;
 * 4: a^-1 (host binary GCD); 5: a^-1 (the kernels' Fermat ladder); 6: a^2 + b^2 through the double product and
 * the squaring kernel; 7: (a - b) * b through the carry-free difference; 8 (nine-limb fields only): 2 (a - b) * y with
 * y the plain value of b, through the NTT butterfly's limb-wise sums, wide carry-free difference and the product by a
 * constant with a precomputed quotient (fx_mul_shoup); 9: 4 (a + b) through limb-wise sums and the lazy reduction. */
int zkt_host_field_op(int curve_id, int which, int op, const uint32_t* a, const uint32_t* b, uint32_t* out);
/* Sum of `count` affine G1 points on the HOST (x, y arkworks Montgomery limbs each; (0, 0) = identity): the combine
 * step of an MSM whose points are sharded across GPUs by index range (the partial sums are
 * all-gathered as raw bytes - no collective can reduce curve points - and added locally).  No context needed. */
int zkt_g1_sum_host(int curve_id, const uint64_t* points_xy_mont, size_t count, uint64_t* out_xy_mont, int* out_is_infinity);
/* Elementwise Fr product on the device (out[i] = a[i]*b[i], Montgomery); test hook for the field
 * kernels. Host pointers. */
int zkt_debug_fr_mul(zkt_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Self-check of the host pairing's shortcuts against their plain definitions (Frobenius maps = powers by p, sparse
 * and cyclotomic products = dense ones, the final exponentiation leaves an element of order r).  0 = all good. */
int zkt_debug_pairing_selftest(int curve_id);
/* The fused quotient pass alone (quotient_poly.rs:98-224), for per-kernel tests: t(x) on the 4n coset from the loaded
 * circuit's ExtendedProverKey and caller-supplied witness cosets.  challenges: alpha beta gamma delta epsilon (5 x 4
 * words); wit: nine HOST vectors of 4n elements in the order a b c pi z1 z2 t h1 h2 (quotient_poly.rs:52-96).  With
 * n_pi > 0 (at most 16) wit[3] is ignored and PI is evaluated the way the prover does for few public inputs, from
 * rotations of the l1 coset (pi_pos: gate indices, pi_vals: 4 words each).  out: 4n elements.  Everything in arkworks
 * Montgomery words.  Whole-coset (single-GPU) circuits only. */
int zkt_debug_quotient(zkt_ctx* ctx, const uint64_t* challenges, const uint64_t* const* wit, const uint64_t* pi_pos,
                       const uint64_t* pi_vals, size_t n_pi, uint64_t* out);
/* The two grand products alone over the loaded circuit's permutation and domain (rows a8 / a9: compute_z1_poly's and
 * compute_z2_poly's evaluation vectors, permutation/mod.rs:181-254, lookup/mod.rs:94-151), through the launches round 3 of
 * the prover makes.  challenges: beta gamma delta epsilon (4 x 4 words); vectors: a b c f t h1 h2, n elements each, host;
 * out_z1 / out_z2: n elements each.  Montgomery words. */
int zkt_debug_grand_products(zkt_ctx* ctx, const uint64_t* challenges, const uint64_t* const* vectors, uint64_t* out_z1,
                             uint64_t* out_z2);
/* kzg10's witness polynomial alone (row a12): out[0 .. len - 1) = (p(X) - p(z)) / (X - z) for the len <= n + 8 coefficients
 * p, as the prover computes it (scaled suffix sums).  Host pointers, Montgomery words. */
int zkt_debug_open_witness(zkt_ctx* ctx, const uint64_t* coeffs, size_t len, const uint64_t* z4, uint64_t* out);
/* Row a13's kernels alone (linearization_poly.rs:55-121): k <= 12 polynomials (lens[j] <= n + 8 coefficients, Montgomery words),
 * each evaluated at points[j] (out_evals: k x 4 words), and the first out_len coefficients of sum_j scalars[j] polys[j]. */
int zkt_debug_eval_lincomb(zkt_ctx* ctx, const uint64_t* const* polys, const size_t* lens, int k, const uint64_t* points,
                           const uint64_t* scalars, uint64_t* out_evals, uint64_t* out_lincomb, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif /* ZKT_PLONK_H */
