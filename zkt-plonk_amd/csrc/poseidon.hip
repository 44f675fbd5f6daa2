// Batched Poseidon over the scalar field: the witness-side arithmetic of the reference's hash gadget (SURVEY.md 8f.3;
// plonk-hashing/src/hasher/poseidon/spec.rs:18-111 rounds, :267-316 round schedule, :239-265 input layout).  One thread
// per hash.  Two kernels:
//   k_poseidon_fx      the permutation (hash values, optionally every round's state): what NativePlonkSpecRef computes;
//   k_poseidon_gadget  the WITNESS of the in-circuit gadget PlonkSpecRef (spec.rs:174-219): every value the composer
//                      assigns while it synthesises one hash -- x^2, x^4, x^5 of each s-box (three mul_gates,
//                      spec.rs:107-111) and each of the W^2 running sums of product_mds (one add_gate per term,
//                      spec.rs:73-88) -- in the composer's allocation order, written into the variable map the prover
//                      gathers its wires from (zkt_prove_inputs.variables).
// Constants are the caller's (the reference generates them at run time, constants.rs:27, or parses the BN254 tables of
// gadgets/src/poseidon).
#include "ctx.hpp"
#include <algorithm>
#include <vector>

#include <cstring>
#include <vector>

namespace zkt {

constexpr int POSEIDON_MAX_WIDTH = 8;

// ---- device-resident form ------------------------------------------------------------------------------------------
// Parameters are uploaded ONCE (zkt_poseidon_load) as 29-bit limbs in the kernels' own Montgomery radix R' = 2^261
// ("H" form, fx.hpp): a product of two H values is an H value, so the whole permutation runs on unpacked limbs with no
// conversion, and sum_i m[i][j] state[i] takes its products two at a time under one reduction (fx_mul2_inl).  Inputs are
// converted on load (one product each), the hash / the optional per-round states on store.
// Lazy bounds of k_poseidon_fx, valid for BOTH fields (R' / p is ~170 on BN254 but only ~71 on BLS12-381, so "state < 9p,
// 81 p^2 < R' p" would NOT do there).  Write rho = R' / p >= 64 (static_assert below).  A product returns a b / R' + p'
// with p' < p.  One operand of every MDS product is a canonical matrix entry (< p), the other a state word < S p, so a
// pair product is < (2 S / rho + 1) p; a state word is the sum of at most four of them plus a round constant:
// S <= 4 (2 S / rho + 1) + 1, i.e. S <= 5 / (1 - 8 / rho) <= 5.72.  The s-box squares a state word: S^2 p^2 <= 32.7 p^2 <
// rho p^2 = R' p, and its later products take operands < 2p and < S p.  fx_mul2's contract (a b + c d < R' p) needs
// 2 S p^2 < rho p^2: 11.5 < 64.
template <class P>
constexpr bool poseidon_lazy_bound_ok() {
    // R' = 2^(29 L) >= 64 p  <=>  p < 2^(29 L - 6): the modulus has at most 29 L - 6 bits
    return P::BITS <= 29 * FxP<P>::L - 6;
}

template <class P>
struct PoseidonFxArgs {
    const uint32_t* rc;    // (2 half_full + partial) * W entries of 9 limbs (H form)
    const uint32_t* mds;   // W * W entries of 9 limbs, m[i][j] at (i * W + j)
    uint32_t tag[FxP<P>::L];
    const Fe<P>* inputs;   // batch * arity, arkworks form, device
    Fe<P>* out;            // batch
    Fe<P>* states;         // optional: batch * (rounds + 1) * W
    uint64_t batch;
    int half_full, partial, arity;
};

template <class P>
ZKT_D Fx<P> fx_load_limbs(const uint32_t* p) {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = p[i];
    return r;
}

template <class P, int W>
__global__ __launch_bounds__(128) void k_poseidon_fx(PoseidonFxArgs<P> a) {
    static_assert(FxP<P>::L == 9, "scalar fields use nine limbs");
    static_assert(poseidon_lazy_bound_ok<P>(), "lazy reduction needs R' >= 64 p (state < 5.72 p, its square < R' p)");
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.batch) return;
    Fx<P> st[W], nx[W];
    // spec.rs:243-265: element 0 = domain tag, the inputs follow, the rest is zero
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) st[0].l[i] = a.tag[i];
#pragma unroll
    for (int i = 1; i < W; ++i)
        st[i] = (i - 1 < a.arity) ? fx_from_ark<P>(fe_load<P>(a.inputs + h * a.arity + (i - 1))) : fx_zero<P>();
    const int rounds = 2 * a.half_full + a.partial;
    Fe<P>* trace = a.states ? a.states + h * (uint64_t)(rounds + 1) * W : nullptr;
    if (trace) {
#pragma unroll
        for (int i = 0; i < W; ++i) fe_store<P>(trace + i, fx_to_ark<P>(st[i]));
    }
    const uint32_t* rc = a.rc;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const bool full = r < a.half_full || r >= a.half_full + a.partial;   // spec.rs:267-316
        // spec.rs:18-37 / 39-71: add the round constants; (x)^5 on every element (full) or on element 0 (partial)
#pragma unroll
        for (int i = 0; i < W; ++i) st[i] = fx_add<P>(st[i], fx_load_limbs<P>(rc + 9 * i));
        rc += 9 * W;
        if (full) {
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const Fx<P> x2 = fx_mul<P>(st[i], st[i]);
                st[i] = fx_mul<P>(fx_mul<P>(x2, x2), st[i]);
            }
        } else {
            const Fx<P> x2 = fx_mul<P>(st[0], st[0]);
            st[0] = fx_mul<P>(fx_mul<P>(x2, x2), st[0]);
        }
        // spec.rs:73-88: result[j] = sum_i m[i][j] * state[i], two products per reduction
#pragma unroll
        for (int j = 0; j < W; ++j) {
            Fx<P> acc;
            bool have = false;
#pragma unroll
            for (int i = 0; i + 1 < W; i += 2) {
                const Fx<P> t = fx_mul2_inl<P>(st[i], fx_load_limbs<P>(a.mds + 9 * (i * W + j)), st[i + 1],
                                               fx_load_limbs<P>(a.mds + 9 * ((i + 1) * W + j)));
                acc = have ? fx_add<P>(acc, t) : t;
                have = true;
            }
            if (W & 1) {
                const Fx<P> t = fx_mul<P>(st[W - 1], fx_load_limbs<P>(a.mds + 9 * ((W - 1) * W + j)));
                acc = have ? fx_add<P>(acc, t) : t;
            }
            nx[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) st[j] = nx[j];
        if (trace) {
#pragma unroll
            for (int i = 0; i < W; ++i) fe_store<P>(trace + (uint64_t)(r + 1) * W + i, fx_to_ark<P>(st[i]));
        }
    }
    fe_store<P>(a.out + h, fx_to_ark<P>(st[1]));   // spec.rs:315: elements[1]
}

// ---- the gadget's witness -------------------------------------------------------------------------------------------
// Everything here stays in arkworks' own Montgomery form ("A": x R, R = 2^256), the form the variable map is stored in,
// so no value is converted on its way out: products against an MDS entry take the entry in H form (A x H -> A), and a
// product of two variables takes one of them shifted up by SH = 5 bits (A x 32 A / R' = A; the shift is two
// instructions per limb, a conversion would be a product).  Every variable is canonical when it is stored, hence
// < p when it is next used: p x 32 p < R' p on both fields.
template <class P>
struct PoseidonGadgetArgs {
    const uint32_t* rc_a;      // round constants as A-form limbs (canonical)
    const uint32_t* mds;       // H form, m[i][j] at (i * W + j)
    uint32_t tag_a[FxP<P>::L];
    const Fe<P>* inputs;       // batch * arity values, or null
    const uint32_t* input_vars;// batch * arity indices into vars, or null
    Fe<P>* vars;               // the variable map
    uint64_t n_vars;
    const uint32_t* trace_base;// per hash, or null: base0 + h * per_hash
    uint64_t base0;
    Fe<P>* out;                // optional: batch hash values
    uint32_t* status;          // set to 1 when a hash was skipped (an index outside the map)
    uint64_t batch;
    int half_full, partial, arity;
};

// value * 2^SH on normalised limbs of a canonical value (32 p < 2^(29 L)); result normalised
template <class P>
ZKT_D Fx<P> fx_shl_sh(const Fx<P>& a) {
    constexpr int L = FxP<P>::L, SH = FxP<P>::SH;
    Fx<P> r;
    r.l[0] = (a.l[0] << SH) & FxP<P>::MASK;
#pragma unroll
    for (int i = 1; i < L - 1; ++i) r.l[i] = ((a.l[i] << SH) | (a.l[i - 1] >> (29 - SH))) & FxP<P>::MASK;
    r.l[L - 1] = (a.l[L - 1] << SH) | (a.l[L - 2] >> (29 - SH));
    return r;
}

// a, b canonical A values -> a b as a canonical A value (mul_gate's assigned value, arithmetic.rs:95-99)
template <class P>
ZKT_D Fx<P> gadget_mul(const Fx<P>& a, const Fx<P>& b_shifted) {
    return fx_cond_sub_p<P>(fx_mul<P>(a, b_shifted));
}

// Stores.  A thread's variables are consecutive in memory, the 64 threads of a wavefront are vars_per_hash x 32 B apart:
// written directly, every store instruction touches 64 different cache lines with 16 bytes each, and the kernel is bound by
// the request rate of that (r04 first version: 1.5-1.7 TB/s of variables, half the issue rate of its own instruction stream).
// So a wavefront stages GADGET_STAGE variables per hash in LDS ([k][lane]: conflict-free writes) and flushes them so that
// 2 GADGET_STAGE consecutive lanes write one hash's GADGET_STAGE x 32 contiguous bytes (two full 128-byte lines at 8):
// 2.4 TB/s on x3 / x4 / x5 alike (4 gives 2.6 / 2.6 / 2.1; profiles/poseidon_r04_v2.txt).
constexpr int GADGET_STAGE = 8;

template <class P, int W>
__global__ __launch_bounds__(128) void k_poseidon_gadget(PoseidonGadgetArgs<P> a) {
    static_assert(FxP<P>::L == 9 && FxP<P>::SH == 5, "scalar fields: nine limbs, R' = 32 R");
    static_assert(P::BITS + FxP<P>::SH < 29 * FxP<P>::L, "p * 2^SH * p < R' p");
    static_assert(sizeof(Fe<P>) == 32, "a variable is two 16-byte words");
    constexpr int K = GADGET_STAGE;
    __shared__ uint4 stage[2][K * 64 * 2];       // per wavefront: [k][lane] x two 16-byte halves
    __shared__ uint64_t sbase[2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = h < a.batch;
    const uint64_t hh = have ? h : 0;            // lanes without a hash run along (barriers below) and write nothing
    const int rounds = 2 * a.half_full + a.partial;
    const uint64_t per_hash = (uint64_t)2 * a.half_full * (3 * W + W * W) + (uint64_t)a.partial * (3 + W * W);
    const uint64_t base = a.trace_base ? (uint64_t)a.trace_base[hh] : a.base0 + hh * per_hash;
    bool ok = base <= a.n_vars && per_hash <= a.n_vars - base;
    Fx<P> st[W], nx[W];
    // reset + input (spec.rs:239-263): LTVariable::constant(domain_tag), the inputs, LTVariable::zero()
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) st[0].l[i] = a.tag_a[i];
#pragma unroll
    for (int i = 1; i < W; ++i) {
        st[i] = fx_zero<P>();
        if (i - 1 < a.arity) {
            if (a.input_vars) {
                const uint32_t v = a.input_vars[hh * a.arity + (i - 1)];
                if (v != ZKT_VARIABLE_ZERO) {
                    if (v < a.n_vars) st[i] = fx_unpack<P>(fe_load<P>(a.vars + v));
                    else ok = false;
                }
            } else {
                st[i] = fx_unpack<P>(fe_load<P>(a.inputs + hh * a.arity + (i - 1)));
            }
        }
    }
    const bool run = have && ok;
    if (have && !ok) atomicOr(a.status, 1u);
    sbase[wv][lane] = run ? base : ~(uint64_t)0;
    uint4* mystage = stage[wv];
    uint32_t cnt = 0;          // variables staged since the last flush (the same in every thread: the schedule is uniform)
    uint64_t pos = 0;          // variables flushed so far
    auto flush = [&](uint32_t k_count) {
        __syncthreads();
        // 16-byte chunk c of the wavefront's k_count x 64 variables: hash c / (2 k_count), then variable, then half
        const uint32_t per = 2u * k_count;
        for (uint32_t c = lane; c < 64u * per; c += 64u) {
            const uint32_t hsh = c / per, part = c - hsh * per, k = part >> 1, half = part & 1u;
            const uint64_t b = sbase[wv][hsh];
            if (b != ~(uint64_t)0)
                reinterpret_cast<uint4*>(a.vars + b + pos + k)[half] = mystage[(k * 64u + hsh) * 2u + half];
        }
        __syncthreads();
        pos += k_count;
        cnt = 0;
    };
    auto emit = [&](const Fx<P>& x) {
        const Fe<P> v = fx_pack<P>(x);
        mystage[(cnt * 64u + lane) * 2u] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
        mystage[(cnt * 64u + lane) * 2u + 1u] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        if (++cnt == (uint32_t)K) flush(K);
    };
    const uint32_t* rc = a.rc_a;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const bool full = r < a.half_full || r >= a.half_full + a.partial;   // output_hash, spec.rs:267-316
        // add_constant (spec.rs:194-200) is a lazy LTVariable transform: no gate, no variable; its value enters the
        // gates that follow
#pragma unroll
        for (int i = 0; i < W; ++i) st[i] = fx_cond_sub_p<P>(fx_add<P>(st[i], fx_load_limbs<P>(rc + 9 * i)));
        rc += 9 * W;
        // power_of_5 (spec.rs:107-111): x^2 = mul(x, x), x^4 = mul(x^2, x^2), x^5 = mul(x^4, x): three variables
#pragma unroll
        for (int i = 0; i < W; ++i) {
            if (i == 0 || full) {
                const Fx<P> xs = fx_shl_sh<P>(st[i]);
                const Fx<P> x2 = gadget_mul<P>(st[i], xs);
                emit(x2);
                const Fx<P> x4 = gadget_mul<P>(x2, fx_shl_sh<P>(x2));
                emit(x4);
                st[i] = gadget_mul<P>(x4, xs);
                emit(st[i]);
            }
        }
        // product_mds (spec.rs:73-88): for j, for i: val = add_gate(val, mul_constant(state[i], m[i][j])): W^2 variables
#pragma unroll
        for (int j = 0; j < W; ++j) {
            Fx<P> acc = fx_zero<P>();
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const Fx<P> t = fx_cond_sub_p<P>(fx_mul<P>(st[i], fx_load_limbs<P>(a.mds + 9 * (i * W + j))));
                acc = fx_cond_sub_p<P>(fx_add<P>(acc, t));
                emit(acc);
            }
            nx[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) st[j] = nx[j];
    }
    if (cnt) flush(cnt);
    if (a.out && run) fe_store<P>(a.out + h, fx_pack<P>(st[1]));   // spec.rs:315: elements[1]
}


// ---- the same witness with W^2 lanes per hash (small batches) --------------------------------------------------------
// One thread per hash walks a dependency chain of 3W + W^2 (full) / 3 + W^2 (partial) products per round: a single
// proof's few hundred hashes (538 in the withdraw circuit at n = 2^20) then take as long as that chain, ~1.7 ms per launch,
// whatever the size of the chip.  Here a hash is spread over LPH >= W^2 lanes of one wavefront: lane (j, i) = j W + i
// holds element i of the state, every lane runs the s-box of its element (three dependent products, redundantly across
// j), multiplies by its own matrix entry m[i][j] (kept in registers for the whole permutation), and the W^2 running sums
// of product_mds are an inclusive scan over i inside each segment j (log2 W shuffle steps).  Depth per round: four products
// instead of 28 - 40, and a round's W^2 sums leave as consecutive 32-byte stores.  Throughput per hash is ~4 x worse than
// k_poseidon_gadget's (lanes idle or redundant in the s-box phase): the host picks this kernel for batches that cannot
// fill the chip with one thread per hash (POSEIDON_LANES_MAX_BATCH).
constexpr uint64_t POSEIDON_LANES_MAX_BATCH = 16384;

template <int W>
struct PoseidonLanes {
    static constexpr int LPH = W * W <= 4 ? 4 : W * W <= 16 ? 16 : W * W <= 32 ? 32 : 64;   // lanes per hash
    static constexpr int PER_WAVE = 64 / LPH;
};

template <class P>
ZKT_D Fx<P> fx_shfl(const Fx<P>& a, int src_lane) {
    Fx<P> r;
#pragma unroll
    for (int i = 0; i < FxP<P>::L; ++i) r.l[i] = (uint32_t)__shfl((int)a.l[i], src_lane);
    return r;
}

template <class P, int W>
__global__ __launch_bounds__(256) void k_poseidon_gadget_lanes(PoseidonGadgetArgs<P> a) {
    static_assert(FxP<P>::L == 9 && FxP<P>::SH == 5, "scalar fields: nine limbs, R' = 32 R");
    constexpr int LPH = PoseidonLanes<W>::LPH, PER_WAVE = PoseidonLanes<W>::PER_WAVE;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPH;                 // which hash of this wavefront
    const int l = lane % LPH;                   // lane inside the hash
    const int seg0 = lane - l;                  // first lane of the hash
    const bool live = l < W * W;
    const int i = live ? l % W : 0, j = live ? l / W : 0;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t h = wave * PER_WAVE + sub;
    const bool have = h < a.batch;              // lanes without a hash run along (shuffles are wave-wide) and store nothing
    const uint64_t hh = have ? h : 0;
    const int rounds = 2 * a.half_full + a.partial;
    const uint64_t per_hash = (uint64_t)2 * a.half_full * (3 * W + W * W) + (uint64_t)a.partial * (3 + W * W);
    const uint64_t base = a.trace_base ? (uint64_t)a.trace_base[hh] : a.base0 + hh * per_hash;
    bool ok = base <= a.n_vars && per_hash <= a.n_vars - base;
    // element i of the initial state (spec.rs:239-263)
    Fx<P> x = fx_zero<P>();
    if (i == 0) {
#pragma unroll
        for (int k = 0; k < FxP<P>::L; ++k) x.l[k] = a.tag_a[k];
    } else if (i - 1 < a.arity) {
        if (a.input_vars) {
            const uint32_t v = a.input_vars[hh * a.arity + (i - 1)];
            if (v != ZKT_VARIABLE_ZERO) {
                if (v < a.n_vars) x = fx_unpack<P>(fe_load<P>(a.vars + v));
                else ok = false;
            }
        } else {
            x = fx_unpack<P>(fe_load<P>(a.inputs + hh * a.arity + (i - 1)));
        }
    }
    // a hash is skipped as a whole: every lane of it must agree
    const unsigned long long bad = __ballot(have && live && !ok);
    const unsigned long long mine = (LPH == 64) ? ~0ull : (((1ull << LPH) - 1ull) << seg0);
    const bool run = have && (bad & mine) == 0;
    if (have && l == 0 && !run) atomicOr(a.status, 1u);
    const Fx<P> m = fx_load_limbs<P>(a.mds + 9 * (i * W + j));    // my matrix entry, H form, for every round
    Fe<P>* out = a.vars + base;
    const uint32_t* rc = a.rc_a + 9 * i;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const bool full = r < a.half_full || r >= a.half_full + a.partial;
        x = fx_cond_sub_p<P>(fx_add<P>(x, fx_load_limbs<P>(rc)));     // add_constant: no gate
        rc += 9 * W;
        // power_of_5 of my element; kept where the round has an s-box for it
        const Fx<P> xs = fx_shl_sh<P>(x);
        const Fx<P> x2 = gadget_mul<P>(x, xs);
        const Fx<P> x4 = gadget_mul<P>(x2, fx_shl_sh<P>(x2));
        const Fx<P> x5 = gadget_mul<P>(x4, xs);
        const bool boxed = full || i == 0;
        if (run && live && j == 0 && boxed) {
            Fe<P>* o = out + 3 * i;
            fe_store<P>(o, fx_pack<P>(x2));
            fe_store<P>(o + 1, fx_pack<P>(x4));
            fe_store<P>(o + 2, fx_pack<P>(x5));
        }
        out += full ? 3 * W : 3;
        Fx<P> s;
#pragma unroll
        for (int k = 0; k < FxP<P>::L; ++k) s.l[k] = boxed ? x5.l[k] : x.l[k];
        // product_mds: my term, then the running sums over i inside segment j (inclusive scan, log2 W steps)
        Fx<P> t = fx_cond_sub_p<P>(fx_mul<P>(s, m));
#pragma unroll
        for (int d = 1; d < W; d <<= 1) {
            const Fx<P> o = fx_shfl<P>(t, lane - d);      // lane - d >= 0 whenever it is used (i >= d)
            if (i >= d) t = fx_cond_sub_p<P>(fx_add<P>(t, o));
        }
        if (run && live) fe_store<P>(out + l, fx_pack<P>(t));
        out += W * W;
        // next state: element i = the last running sum of segment i
        x = fx_shfl<P>(t, seg0 + i * W + (W - 1));
    }
    if (a.out && run && l == 1 % W) {   // spec.rs:315: elements[1] (W >= 2: lane 1 holds element 1)
        fe_store<P>(a.out + h, fx_pack<P>(x));
    }
}

}  // namespace zkt

// the opaque handle of include/zkt_plonk.h: PoseidonConstants resident in HBM
struct zkt_poseidon {
    int curve = 0, width = 0, half_full = 0, partial = 0;
    void* d_rc = nullptr;    // limbs, H form
    void* d_mds = nullptr;
    void* d_rc_a = nullptr;  // limbs, arkworks form (the gadget kernel's)
    void* d_status = nullptr;// one word: a gadget launch skipped a hash (index outside the variable map)
    uint32_t tag[16] = {};
    uint32_t tag_a[16] = {};
};

namespace zkt {

template <class P>
static void to_h_limbs(const uint64_t* src_mont, size_t count, std::vector<uint32_t>& out) {
    out.resize(count * 9);
    for (size_t k = 0; k < count; ++k) {
        Fe<P> v;
        memcpy(v.v, src_mont + 4 * k, 32);
        const Fx<P> x = fx_cond_sub_p<P>(fx_from_ark<P>(v));   // canonical H form
        for (int i = 0; i < 9; ++i) out[9 * k + i] = x.l[i];
    }
}

// arkworks-form words -> canonical A-form limbs (no conversion: the same residue x R, unpacked)
template <class P>
static int to_a_limbs(const uint64_t* src_mont, size_t count, std::vector<uint32_t>& out) {
    out.resize(count * 9);
    for (size_t k = 0; k < count; ++k) {
        Fe<P> v;
        memcpy(v.v, src_mont + 4 * k, 32);
        const Fx<P> x = fx_unpack<P>(v);
        const Fx<P> cx = fx_cond_sub_p<P>(x);
        for (int i = 0; i < 9; ++i) {
            if (cx.l[i] != x.l[i]) return 1;   // not below the modulus
            out[9 * k + i] = x.l[i];
        }
    }
    return 0;
}

template <class P>
static int poseidon_load_t(zkt_ctx* c, const zkt_poseidon_params& p, zkt_poseidon* h) {
    const int W = p.width, rounds = 2 * p.half_full_rounds + p.partial_rounds;
    std::vector<uint32_t> rc, mds, tag, rc_a, tag_a;
    to_h_limbs<P>(p.round_constants, (size_t)rounds * W, rc);
    to_h_limbs<P>(p.mds, (size_t)W * W, mds);
    to_h_limbs<P>(p.domain_tag, 1, tag);
    if (to_a_limbs<P>(p.round_constants, (size_t)rounds * W, rc_a) || to_a_limbs<P>(p.domain_tag, 1, tag_a))
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: a constant is not below the field modulus");
    for (int i = 0; i < 9; ++i) h->tag[i] = tag[i];
    for (int i = 0; i < 9; ++i) h->tag_a[i] = tag_a[i];
    int rcode;
    if ((rcode = dev_alloc(c, &h->d_rc, rc.size() * 4))) return rcode;
    if ((rcode = dev_alloc(c, &h->d_mds, mds.size() * 4))) return rcode;
    if ((rcode = dev_alloc(c, &h->d_rc_a, rc_a.size() * 4))) return rcode;
    if ((rcode = dev_alloc(c, &h->d_status, 4))) return rcode;
    ZKT_HIP(c, hipMemsetAsync(h->d_status, 0, 4, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(h->d_rc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(h->d_mds, mds.data(), mds.size() * 4, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(h->d_rc_a, rc_a.data(), rc_a.size() * 4, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));   // the staging vectors die with this call
    return ZKT_OK;
}

template <class P, int W>
static void poseidon_launch_w(zkt_ctx* c, const PoseidonFxArgs<P>& a) {
    hipLaunchKernelGGL((k_poseidon_fx<P, W>), dim3((unsigned)((a.batch + 127) / 128)), dim3(128), 0, c->stream, a);
}

template <class P>
static int poseidon_enqueue_t(zkt_ctx* c, const zkt_poseidon* h, const void* d_inputs, size_t batch, int arity, void* d_out,
                              void* d_states) {
    PoseidonFxArgs<P> a{};
    a.rc = (const uint32_t*)h->d_rc;
    a.mds = (const uint32_t*)h->d_mds;
    for (int i = 0; i < 9; ++i) a.tag[i] = h->tag[i];
    a.inputs = (const Fe<P>*)d_inputs;
    a.out = (Fe<P>*)d_out;
    a.states = (Fe<P>*)d_states;
    a.batch = batch;
    a.half_full = h->half_full;
    a.partial = h->partial;
    a.arity = arity;
    switch (h->width) {
        case 2: poseidon_launch_w<P, 2>(c, a); break;
        case 3: poseidon_launch_w<P, 3>(c, a); break;
        case 4: poseidon_launch_w<P, 4>(c, a); break;
        case 5: poseidon_launch_w<P, 5>(c, a); break;
        case 6: poseidon_launch_w<P, 6>(c, a); break;
        case 7: poseidon_launch_w<P, 7>(c, a); break;
        default: poseidon_launch_w<P, 8>(c, a); break;
    }
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}

template <class P, int W>
static void poseidon_gadget_launch_w(zkt_ctx* c, const PoseidonGadgetArgs<P>& a, int mode) {
    const bool lanes = mode == 2 || (mode == 0 && a.batch <= POSEIDON_LANES_MAX_BATCH);
    if (lanes) {
        constexpr uint64_t per_block = 4 * PoseidonLanes<W>::PER_WAVE;    // 256 threads = 4 wavefronts
        hipLaunchKernelGGL((k_poseidon_gadget_lanes<P, W>), dim3((unsigned)((a.batch + per_block - 1) / per_block)), dim3(256), 0,
                           c->stream, a);
    } else {
        hipLaunchKernelGGL((k_poseidon_gadget<P, W>), dim3((unsigned)((a.batch + 127) / 128)), dim3(128), 0, c->stream, a);
    }
}

template <class P>
static int poseidon_gadget_enqueue_t(zkt_ctx* c, const zkt_poseidon* h, const zkt_poseidon_gadget_args& g) {
    PoseidonGadgetArgs<P> a{};
    a.rc_a = (const uint32_t*)h->d_rc_a;
    a.mds = (const uint32_t*)h->d_mds;
    for (int i = 0; i < 9; ++i) a.tag_a[i] = h->tag_a[i];
    a.inputs = (const Fe<P>*)g.d_inputs;
    a.input_vars = g.d_input_vars;
    a.vars = (Fe<P>*)g.d_variables;
    a.n_vars = g.n_vars;
    a.trace_base = g.d_trace_base;
    a.base0 = g.trace_base0;
    a.out = (Fe<P>*)g.d_out_hashes;
    a.status = (uint32_t*)h->d_status;
    a.batch = g.batch;
    a.half_full = h->half_full;
    a.partial = h->partial;
    a.arity = g.arity;
    const int mode = g.kernel;   // 0: by batch size; 1: one thread per hash; 2: W^2 lanes per hash
    switch (h->width) {
        case 2: poseidon_gadget_launch_w<P, 2>(c, a, mode); break;
        case 3: poseidon_gadget_launch_w<P, 3>(c, a, mode); break;
        case 4: poseidon_gadget_launch_w<P, 4>(c, a, mode); break;
        case 5: poseidon_gadget_launch_w<P, 5>(c, a, mode); break;
        case 6: poseidon_gadget_launch_w<P, 6>(c, a, mode); break;
        case 7: poseidon_gadget_launch_w<P, 7>(c, a, mode); break;
        default: poseidon_gadget_launch_w<P, 8>(c, a, mode); break;
    }
    ZKT_HIP(c, hipGetLastError());
    return ZKT_OK;
}

static int poseidon_check_params(zkt_ctx* c, const zkt_poseidon_params* p) {
    if (!p || !p->round_constants || !p->mds || !p->domain_tag) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    // output_hash (spec.rs:267-316) always runs one full and one partial round before its `1..n` loops: a schedule with
    // no partial round does not exist in the reference
    if (p->width < 2 || p->width > POSEIDON_MAX_WIDTH || p->half_full_rounds < 1 || p->partial_rounds < 1)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: width in [2, 8], half_full_rounds >= 1, partial_rounds >= 1");
    return ZKT_OK;
}

}  // namespace zkt

using namespace zkt;

extern "C" {

int zkt_poseidon_load(zkt_ctx* c, const zkt_poseidon_params* p, zkt_poseidon** out) {
    if (!c || !out) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (int rc = poseidon_check_params(c, p)) return rc;
    (void)hipSetDevice(c->device);
    zkt_poseidon* h = new zkt_poseidon();
    h->curve = c->curve;
    h->width = p->width;
    h->half_full = p->half_full_rounds;
    h->partial = p->partial_rounds;
    const int rc = c->curve == ZKT_CURVE_BN254 ? poseidon_load_t<Bn254Fr>(c, *p, h) : poseidon_load_t<Bls381Fr>(c, *p, h);
    if (rc) {
        dev_free(c, h->d_rc);
        dev_free(c, h->d_mds);
        dev_free(c, h->d_rc_a);
        dev_free(c, h->d_status);
        delete h;
        return rc;
    }
    *out = h;
    return ZKT_OK;
}

void zkt_poseidon_free(zkt_ctx* c, zkt_poseidon* h) {
    if (!h) return;
    if (c) {
        (void)hipStreamSynchronize(c->stream);
        dev_free(c, h->d_rc);
        dev_free(c, h->d_mds);
        dev_free(c, h->d_rc_a);
        dev_free(c, h->d_status);
    }
    delete h;
}

int zkt_poseidon_hash_batch_dev(zkt_ctx* c, const zkt_poseidon* h, const void* d_inputs, size_t batch, int arity,
                                void* d_out_hashes, void* d_out_states) {
    if (!c || !h || !d_out_hashes || (!d_inputs && batch && arity)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (h->curve != c->curve) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon parameters belong to another curve");
    if (arity < 0 || arity > h->width - 1)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: arity <= width - 1 (spec.rs:253-257 FullBuffer)");
    if (batch == 0) return ZKT_OK;
    (void)hipSetDevice(c->device);
    if (c->curve == ZKT_CURVE_BN254) return poseidon_enqueue_t<Bn254Fr>(c, h, d_inputs, batch, arity, d_out_hashes, d_out_states);
    return poseidon_enqueue_t<Bls381Fr>(c, h, d_inputs, batch, arity, d_out_hashes, d_out_states);
}

size_t zkt_poseidon_gadget_vars_per_hash(const zkt_poseidon* h) {
    if (!h) return 0;
    const size_t W = (size_t)h->width;
    return (size_t)2 * h->half_full * (3 * W + W * W) + (size_t)h->partial * (3 + W * W);
}

int zkt_poseidon_gadget_witness_dev(zkt_ctx* c, const zkt_poseidon* h, const zkt_poseidon_gadget_args* g) {
    if (!c || !h || !g) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (h->curve != c->curve) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon parameters belong to another curve");
    if (g->arity < 0 || g->arity > h->width - 1)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: arity <= width - 1 (spec.rs:253-257 FullBuffer)");
    if (g->batch == 0) return ZKT_OK;
    if (!g->d_variables) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null variable map");
    if (g->arity && !g->d_inputs == !g->d_input_vars)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: exactly one of d_inputs / d_input_vars");
    if (g->n_vars > 0xFFFFFFFFull) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "variable indices are 32 bits wide");
    if (g->kernel < 0 || g->kernel > 2) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: kernel = 0 (auto), 1 or 2");
    const size_t per = zkt_poseidon_gadget_vars_per_hash(h);
    if (!g->d_trace_base && (g->trace_base0 > g->n_vars || g->batch > (g->n_vars - g->trace_base0) / per))
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: the traces do not fit the variable map");
    (void)hipSetDevice(c->device);
    if (c->curve == ZKT_CURVE_BN254) return poseidon_gadget_enqueue_t<Bn254Fr>(c, h, *g);
    return poseidon_gadget_enqueue_t<Bls381Fr>(c, h, *g);
}

// Debug validation of one launch's structure (host side; synchronises): the traces [base, base + vars_per_hash) must be
// pairwise disjoint and inside the map, and no input index may lie inside a trace of the SAME launch (the kernel's hashes
// are independent: such an input would be read before, while or after it is written).  A caller that schedules its own
// launches (the Python mirror does it in PoseidonGadget.levels) runs this once per circuit, not per proof.
int zkt_poseidon_gadget_validate(zkt_ctx* c, const zkt_poseidon* h, const zkt_poseidon_gadget_args* g) {
    if (!c || !h || !g) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (g->arity < 0 || g->arity > h->width - 1)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: arity <= width - 1 (spec.rs:253-257 FullBuffer)");
    if (g->batch == 0) return ZKT_OK;
    (void)hipSetDevice(c->device);
    const size_t per = zkt_poseidon_gadget_vars_per_hash(h);
    std::vector<uint64_t> base(g->batch);
    if (g->d_trace_base) {
        std::vector<uint32_t> b32(g->batch);
        ZKT_HIP(c, hipMemcpyAsync(b32.data(), g->d_trace_base, g->batch * 4, hipMemcpyDeviceToHost, c->stream));
        ZKT_HIP(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < g->batch; ++i) base[i] = b32[i];
    } else {
        for (size_t i = 0; i < g->batch; ++i) base[i] = g->trace_base0 + i * per;
    }
    std::vector<uint64_t> sorted(base);
    std::sort(sorted.begin(), sorted.end());
    for (size_t i = 0; i < sorted.size(); ++i) {
        if (sorted[i] > g->n_vars || per > g->n_vars - sorted[i])
            return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: a trace lies outside the variable map");
        if (i && sorted[i] < sorted[i - 1] + per)
            return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: two traces of one launch overlap");
    }
    if (g->arity && g->d_input_vars) {
        std::vector<uint32_t> idx(g->batch * (size_t)g->arity);
        ZKT_HIP(c, hipMemcpyAsync(idx.data(), g->d_input_vars, idx.size() * 4, hipMemcpyDeviceToHost, c->stream));
        ZKT_HIP(c, hipStreamSynchronize(c->stream));
        for (uint32_t v : idx) {
            if (v == 0xFFFFFFFFu) continue;   // Variable::Zero
            if (v >= g->n_vars) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: an input index lies outside the variable map");
            auto it = std::upper_bound(sorted.begin(), sorted.end(), (uint64_t)v);
            if (it != sorted.begin() && (uint64_t)v < *(it - 1) + per)
                return set_err(c, ZKT_ERR_INVALID_ARGUMENT,
                               "poseidon gadget: an input is a variable the same launch writes (run it in a later launch)");
        }
    }
    return ZKT_OK;
}

int zkt_poseidon_gadget_check(zkt_ctx* c, const zkt_poseidon* h) {
    if (!c || !h) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    uint32_t st = 0;
    ZKT_HIP(c, hipMemcpyAsync(&st, h->d_status, 4, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    if (st) {
        ZKT_HIP(c, hipMemsetAsync(h->d_status, 0, 4, c->stream));
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon gadget: a trace base or an input index lies outside the variable map");
    }
    return ZKT_OK;
}

// host-pointer convenience form: load, stage, run, download, release (every exit path frees what it took)
int zkt_poseidon_hash_batch(zkt_ctx* c, const zkt_poseidon_params* p, const uint64_t* inputs, size_t batch, int arity,
                            uint64_t* out_hashes, uint64_t* out_states) {
    if (!c || !out_hashes || (!inputs && batch && arity)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (int rc0 = poseidon_check_params(c, p)) return rc0;
    if (arity < 0 || arity > p->width - 1)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: width in [2, 8], arity <= width - 1 (spec.rs:253-257 FullBuffer)");
    if (batch == 0) return ZKT_OK;
    (void)hipSetDevice(c->device);
    const int W = p->width, rounds = 2 * p->half_full_rounds + p->partial_rounds;
    const size_t n_in = batch * (size_t)arity, n_st = out_states ? batch * (size_t)(rounds + 1) * W : 0;
    zkt_poseidon* h = nullptr;
    void *d_in = nullptr, *d_out = nullptr, *d_st = nullptr;
    auto run = [&]() -> int {
        int rc;
        if ((rc = zkt_poseidon_load(c, p, &h))) return rc;
        if ((rc = dev_alloc(c, &d_in, (n_in ? n_in : 1) * 32))) return rc;
        if ((rc = dev_alloc(c, &d_out, batch * 32))) return rc;
        if (n_st && (rc = dev_alloc(c, &d_st, n_st * 32))) return rc;
        if (n_in) ZKT_HIP(c, hipMemcpyAsync(d_in, inputs, n_in * 32, hipMemcpyHostToDevice, c->stream));
        if ((rc = zkt_poseidon_hash_batch_dev(c, h, d_in, batch, arity, d_out, d_st))) return rc;
        ZKT_HIP(c, hipMemcpyAsync(out_hashes, d_out, batch * 32, hipMemcpyDeviceToHost, c->stream));
        if (n_st) ZKT_HIP(c, hipMemcpyAsync(out_states, d_st, n_st * 32, hipMemcpyDeviceToHost, c->stream));
        ZKT_HIP(c, hipStreamSynchronize(c->stream));
        return ZKT_OK;
    };
    const int rc = run();
    if (rc) (void)hipStreamSynchronize(c->stream);
    dev_free(c, d_in);
    dev_free(c, d_out);
    dev_free(c, d_st);
    zkt_poseidon_free(c, h);
    return rc;
}

}  // extern "C"
