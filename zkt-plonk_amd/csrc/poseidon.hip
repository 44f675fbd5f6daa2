// Batched Poseidon permutation over the scalar field: the witness-side arithmetic of the reference's hash gadget
// (plonk-hashing/src/hasher/poseidon/spec.rs:18-111 rounds, :267-316 round schedule, :239-265 input
// layout).  One thread per hash; with `out_states` every round's state is kept, which is what a composer needs to
// fill the variables of its Poseidon gates for a whole batch of Merkle paths / notes at once.
// Constants are the caller's (the reference generates them at run time, constants.rs:27: there is no table to pin).
#include "ctx.hpp"

#include <cstring>
#include <vector>

namespace zkt {

constexpr int POSEIDON_MAX_WIDTH = 8;

// ---- device-resident form ------------------------------------------------------------------------------------------
// Parameters are uploaded ONCE (zkt_poseidon_load) as 29-bit limbs in the kernels' own Montgomery radix R' = 2^261
// ("H" form, fx.hpp): a product of two H values is an H value, so the whole permutation runs on unpacked limbs with no
// conversion, and sum_i m[i][j] state[i] takes its products two at a time under one reduction (fx_mul2_inl).  Inputs are
// converted on load (one product each), the hash / the optional per-round states on store.  Lazy bounds: a state word
// is a sum of at most four products (< 8p), plus a round constant (< 9p); 81 p^2 < R' p, so no step needs a reduction.
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

}  // namespace zkt

// the opaque handle of include/zkt_plonk.h: PoseidonConstants resident in HBM
struct zkt_poseidon {
    int curve = 0, width = 0, half_full = 0, partial = 0;
    void* d_rc = nullptr;    // limbs, H form
    void* d_mds = nullptr;
    uint32_t tag[16] = {};
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

template <class P>
static int poseidon_load_t(zkt_ctx* c, const zkt_poseidon_params& p, zkt_poseidon* h) {
    const int W = p.width, rounds = 2 * p.half_full_rounds + p.partial_rounds;
    std::vector<uint32_t> rc, mds, tag;
    to_h_limbs<P>(p.round_constants, (size_t)rounds * W, rc);
    to_h_limbs<P>(p.mds, (size_t)W * W, mds);
    to_h_limbs<P>(p.domain_tag, 1, tag);
    for (int i = 0; i < 9; ++i) h->tag[i] = tag[i];
    int rcode;
    if ((rcode = dev_alloc(c, &h->d_rc, rc.size() * 4))) return rcode;
    if ((rcode = dev_alloc(c, &h->d_mds, mds.size() * 4))) return rcode;
    ZKT_HIP(c, hipMemcpyAsync(h->d_rc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(h->d_mds, mds.data(), mds.size() * 4, hipMemcpyHostToDevice, c->stream));
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
