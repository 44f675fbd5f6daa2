// Batched Poseidon permutation over the scalar field: the witness-side arithmetic of the reference's hash gadget
// (plonk-hashing/src/hasher/poseidon/spec.rs:18-111 rounds, :267-316 round schedule, :239-265 input
// layout).  One thread per hash; with `out_states` every round's state is kept, which is what a composer needs to
// fill the variables of its Poseidon gates for a whole batch of Merkle paths / notes at once.
// Constants are the caller's (the reference generates them at run time, constants.rs:27: there is no table to pin).
#include "ctx.hpp"

#include <cstring>

namespace zkt {

constexpr int POSEIDON_MAX_WIDTH = 8;

template <class P>
struct PoseidonDev {
    const Fe<P>* rc;      // (2 * half_full + partial) * width
    const Fe<P>* mds;     // width * width, m[i][j] at i * width + j
    Fe<P> domain_tag;
    const Fe<P>* inputs;  // batch * arity
    Fe<P>* out;           // batch
    Fe<P>* states;        // optional: batch * (rounds + 1) * width
    uint64_t batch;
    int width, half_full, partial, arity;
};

template <class P>
ZKT_D Fe<P> pow5(const Fe<P>& x) {   // spec.rs:108-112
    const Fe<P> x2 = fe_sqr<P>(x), x4 = fe_sqr<P>(x2);
    return fe_mul<P>(x4, x);
}

template <class P>
__global__ __launch_bounds__(128) void k_poseidon(PoseidonDev<P> a) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.batch) return;
    const int W = a.width;
    Fe<P> st[POSEIDON_MAX_WIDTH], nx[POSEIDON_MAX_WIDTH];
    // spec.rs:243-265: element 0 = domain tag, the inputs follow, the rest is zero
    st[0] = a.domain_tag;
    for (int i = 1; i < W; ++i) st[i] = (i - 1 < a.arity) ? fe_load<P>(a.inputs + h * a.arity + (i - 1)) : fe_zero<P>();
    const int rounds = 2 * a.half_full + a.partial;
    Fe<P>* trace = a.states ? a.states + h * (uint64_t)(rounds + 1) * W : nullptr;
    if (trace) for (int i = 0; i < W; ++i) fe_store<P>(trace + i, st[i]);
    int off = 0;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const bool full = r < a.half_full || r >= a.half_full + a.partial;   // spec.rs:267-316
        if (full) {   // spec.rs:18-37: (x + rc)^5 on every element
            for (int i = 0; i < W; ++i) st[i] = pow5<P>(fe_add<P>(st[i], fe_load<P>(a.rc + off + i)));
        } else {      // spec.rs:39-54: add the round constants, s-box on element 0 only
            for (int i = 0; i < W; ++i) st[i] = fe_add<P>(st[i], fe_load<P>(a.rc + off + i));
            st[0] = pow5<P>(st[0]);
        }
        off += W;
        // spec.rs:73-88: result[j] = sum_i m[i][j] * state[i]
        for (int j = 0; j < W; ++j) {
            Fe<P> acc = fe_zero<P>();
            for (int i = 0; i < W; ++i) acc = fe_add<P>(acc, fe_mul<P>(st[i], fe_load<P>(a.mds + i * W + j)));
            nx[j] = acc;
        }
        for (int j = 0; j < W; ++j) st[j] = nx[j];
        if (trace) for (int i = 0; i < W; ++i) fe_store<P>(trace + (uint64_t)(r + 1) * W + i, st[i]);
    }
    fe_store<P>(a.out + h, st[1]);   // spec.rs:315: elements[1]
}

template <class P>
static int poseidon_t(zkt_ctx* c, const zkt_poseidon_params& p, const uint64_t* inputs, size_t batch, int arity, uint64_t* out,
                      uint64_t* out_states) {
    const int W = p.width, rounds = 2 * p.half_full_rounds + p.partial_rounds;
    const size_t n_rc = (size_t)rounds * W, n_in = batch * (size_t)arity, n_st = out_states ? batch * (size_t)(rounds + 1) * W : 0;
    void *d_rc = nullptr, *d_mds = nullptr, *d_in = nullptr, *d_out = nullptr, *d_st = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_rc, n_rc * 32))) return rc;
    if ((rc = dev_alloc(c, &d_mds, (size_t)W * W * 32))) return rc;
    if ((rc = dev_alloc(c, &d_in, (n_in ? n_in : 1) * 32))) return rc;
    if ((rc = dev_alloc(c, &d_out, batch * 32))) return rc;
    if (n_st && (rc = dev_alloc(c, &d_st, n_st * 32))) return rc;
    ZKT_HIP(c, hipMemcpyAsync(d_rc, p.round_constants, n_rc * 32, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(d_mds, p.mds, (size_t)W * W * 32, hipMemcpyHostToDevice, c->stream));
    if (n_in) ZKT_HIP(c, hipMemcpyAsync(d_in, inputs, n_in * 32, hipMemcpyHostToDevice, c->stream));
    PoseidonDev<P> a{};
    a.rc = (const Fe<P>*)d_rc;
    a.mds = (const Fe<P>*)d_mds;
    memcpy(a.domain_tag.v, p.domain_tag, 32);
    a.inputs = (const Fe<P>*)d_in;
    a.out = (Fe<P>*)d_out;
    a.states = (Fe<P>*)d_st;
    a.batch = batch;
    a.width = W;
    a.half_full = p.half_full_rounds;
    a.partial = p.partial_rounds;
    a.arity = arity;
    hipLaunchKernelGGL(k_poseidon<P>, dim3((unsigned)((batch + 127) / 128)), dim3(128), 0, c->stream, a);
    ZKT_HIP(c, hipGetLastError());
    ZKT_HIP(c, hipMemcpyAsync(out, d_out, batch * 32, hipMemcpyDeviceToHost, c->stream));
    if (n_st) ZKT_HIP(c, hipMemcpyAsync(out_states, d_st, n_st * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    dev_free(c, d_rc); dev_free(c, d_mds); dev_free(c, d_in); dev_free(c, d_out); dev_free(c, d_st);
    return ZKT_OK;
}

}  // namespace zkt

using namespace zkt;

extern "C" int zkt_poseidon_hash_batch(zkt_ctx* c, const zkt_poseidon_params* p, const uint64_t* inputs, size_t batch, int arity,
                                       uint64_t* out_hashes, uint64_t* out_states) {
    if (!c || !p || !out_hashes || (!inputs && batch && arity) || !p->round_constants || !p->mds || !p->domain_tag)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (p->width < 2 || p->width > POSEIDON_MAX_WIDTH || arity < 0 || arity > p->width - 1 || p->half_full_rounds < 1 ||
        p->partial_rounds < 0)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "poseidon: width in [2, 8], arity <= width - 1 (spec.rs:253-257 FullBuffer)");
    if (batch == 0) return ZKT_OK;
    (void)hipSetDevice(c->device);
    if (c->curve == ZKT_CURVE_BN254) return poseidon_t<Bn254Fr>(c, *p, inputs, batch, arity, out_hashes, out_states);
    return poseidon_t<Bls381Fr>(c, *p, inputs, batch, arity, out_hashes, out_states);
}
