// Device-side polynomial / evaluation-vector kernels of the prover rounds (launch wrappers).
// Each wrapper cites the reference code it replaces; implementations are in poly.hip.
#pragma once
#include "ctx.hpp"

namespace zkt {

constexpr int LC_MAX_TERMS = 16;
struct LinCombArgs {            // out[i] = sum_k scalar[k] * poly[k][i]  (poly k read as zero beyond len[k])
    const void* poly[LC_MAX_TERMS];
    uint64_t len[LC_MAX_TERMS];
    uint32_t scalar[LC_MAX_TERMS][8];  // Montgomery
    int nterms;
};

constexpr int EVAL_MAX = 16;
struct EvalArgs {               // result[k] = poly[k](point[k])
    const void* poly[EVAL_MAX];
    uint64_t len[EVAL_MAX];
    uint32_t point[EVAL_MAX][8];
    int count;
};

struct QuotientArgs {           // all vectors hold 4n coset evaluations
    const void *a, *b, *c, *pi, *z1, *z2, *t, *h1, *h2;
    const void *q_m, *q_l, *q_r, *q_o, *q_c, *q_lookup, *q_table, *sigma1, *sigma2, *sigma3, *x, *l1;
    void* out;
    uint32_t alpha[8], beta[8], gamma[8], delta[8], epsilon[8];
    uint32_t zh_inv[4][8];      // 1 / (x^n - 1) on the coset: depends on i mod 4 only
    uint64_t n4;
    // Few public inputs: pi is not transformed at all.  PI(X) = sum_i v_i L_{pos_i}(X) and L_j(x) = L_0(x w^-j), so on
    // the 4n coset PI[k] = sum_i v_i * l1[(k - 4 pos_i) mod 4n].  pi_tab (device): n_pi_direct entries of 10 words,
    // {4 pos_i, nine 29-bit limbs of v_i (arkworks Montgomery form)}; `pi` is ignored when pi_tab is set.
    const uint32_t* pi_tab;
    uint32_t n_pi_direct;
    // Class form (one GPU's share of the 4n coset in a proof sharded over G GPUs, include/zkt_plonk.h): the vectors hold
    // the n4 = 4n / G points of global index cls + G i.  "omega-next" (global index + 4) of point i is entry
    // (i + next_off) mod n4 of the *_next vectors: the same vectors with next_off = 4 / G for G <= 4, the neighbouring
    // class (cls + 4) mod 8 with next_off = (cls + 4) / 8 for G = 8.  zh_inv is indexed by the global index mod 4; the
    // rotations in pi_tab are in class entries (4 pos / G).  G = 0 means the whole coset (G = 1, cls = 0, next_off = 4).
    uint32_t G, cls, next_off;
    const void *z1_next, *z2_next, *t_next, *h1_next;
    // only the points [first, first + count) of the vectors (count = 0: all n4); out[i] is written for those i
    uint64_t first, count;
};
constexpr int QUOTIENT_PI_DIRECT_MAX = 16;

struct ZTermsArgs {
    const void *a, *b, *c, *s1, *s2, *s3, *roots;  // z1
    const void *f, *t, *h1, *h2;                    // z2
    void *num, *den;
    uint32_t beta[8], gamma[8], delta[8], epsilon[8];
    uint64_t n;
};

// elementwise
int poly_mul_vec(zkt_ctx* c, const void* a, const void* b, void* out, size_t n);              // prove.rs:157-161
int poly_set_zero(zkt_ctx* c, void* p, size_t n_elems);
// from_coefficients_vec: *d_len (zero on entry) = number of coefficients after stripping trailing zeros of p[0..n);
// optionally clears zero_count elements at zero_at in the same launch (the slack above a transform's output)
int poly_trim_len(zkt_ctx* c, const void* p, size_t n, uint32_t* d_len, void* zero_at = nullptr, int zero_count = 0);
int poly_copy_pad(zkt_ctx* c, const void* d_in, size_t len, void* out, size_t n);   // prove.rs:39-55 pad_to
// prove.rs:49-55 wire_evals: out[i] = values[idx[i]] (0xFFFFFFFF = Variable::Zero), zero padded to n; d_status |= 16 on
// an index >= n_vars
int poly_gather_pad(zkt_ctx* c, const void* d_values, size_t n_vars, const uint32_t* d_idx, size_t rows, void* out, size_t n,
                    uint32_t* d_status);
int poly_add_blinders(zkt_ctx* c, void* p, const uint32_t* d_len, const void* d_blinders, int k, size_t cap);  // prove.rs:472-483
int poly_lincomb(zkt_ctx* c, const LinCombArgs& a, void* out, size_t n);
// scalars (n + k of them) of a commitment taken against the Lagrange-prefix table (lagrange.hip) for the polynomial with
// evaluations ev[0..n) and, when k > 0, the k blinders of prove.rs:472-483 appended at its trimmed length *d_len
int lagrange_scalars(zkt_ctx* c, const void* ev, size_t n, const uint32_t* d_len, const void* d_blinders, int k, const void* d_roots,
                     void* out);
// d_powers: scratch of EVAL_MAX * (257 + ceil(maxlen / 2048)) elements (powers of each evaluation point)
int poly_eval_many(zkt_ctx* c, const EvalArgs& a, void* d_partials, void* d_results, void* d_powers);         // linearization_poly.rs:55-75
// grand products (permutation/mod.rs:181-254, lookup/mod.rs:94-151)
int z1_terms(zkt_ctx* c, const ZTermsArgs& a);
int z2_terms(zkt_ctx* c, const ZTermsArgs& a);
int scan_mul(zkt_ctx* c, const void* in, void* out, size_t n, bool reverse, void* d_tmp /* >= 2*(n/1024 + 2048) elems */);
int scan_add(zkt_ctx* c, const void* in, void* out, size_t n, bool reverse, void* d_tmp);
int z_combine(zkt_ctx* c, const void* pn, const void* sd, const uint32_t inv_total[8], void* out, size_t n);
// quotient (quotient_poly.rs:98-224)
int quotient_pointwise(zkt_ctx* c, const QuotientArgs& a);
// in place: arkworks Montgomery form -> the quotient kernel's R' = 2^261 form times 32^k32 (k32 in {0, 1}).
// quotient_pointwise expects q_l q_r q_o q_lookup q_table l1 with k32 = 0 and q_m with k32 = 1.
int to_hat_form(zkt_ctx* c, void* v, size_t n, int k32);
// class-major (rank r's n4 / G points at [r * n4 / G, ...)) -> natural order of the 4n coset: out[cls + G i] = in[cls * (n4 / G) + i].
// chunks > 1: the exchange went out in pieces, `in` is [chunk][class][n4 / G / chunks]
int quotient_interleave(zkt_ctx* c, const void* in, void* out, size_t n4, uint32_t G, uint32_t chunks = 1);
int quotient_split_blind(zkt_ctx* c, const void* q, size_t n, const void* d_b0b1, void* q_lo, void* q_mid, void* q_hi,
                         uint32_t* d_status);                                                   // prove.rs:287-300
// KZG opening witness: w = p / (X - z)  (kzg10::compute_witness_polynomial)
int open_witness(zkt_ctx* c, const void* p, size_t len, const uint32_t z[8], const uint32_t z_inv[8], void* d_tmp_a, void* d_tmp_b,
                 void* d_scan_tmp, void* out, void* d_powers /* open_witness_powers(len) elements of scratch */);
size_t open_witness_powers(size_t maxlen);   // elements: the powers of z and 1/z the two scalings read
// table generation
int gen_powers(zkt_ctx* c, void* out, size_t n, const uint32_t base[8], const uint32_t scale[8]);
// Plookup sorted halves h1/h2 (lookup/multiset.rs:103-146)
int lookup_count(zkt_ctx* c, const void* f, size_t n, const void* d_sorted_keys, const uint32_t* d_perm, uint32_t nkeys,
                 uint32_t* d_counts, uint32_t* d_status);
// start offsets of both halves of combine_split (multiset.rs:126-143) from the per-key counts base + hits; nkeys + 1
// entries each.  d_hits is cleared on the way.  d_status |= 8 when a half does not come out at n elements.
int lookup_starts(zkt_ctx* c, const uint32_t* d_base_counts, uint32_t* d_hits, uint32_t nkeys, size_t n, uint32_t* d_even,
                  uint32_t* d_odd, uint32_t* d_status);
int lookup_expand(zkt_ctx* c, const void* d_keys_insertion, const uint32_t* d_starts, uint32_t nkeys, void* out, size_t n);

}  // namespace zkt
