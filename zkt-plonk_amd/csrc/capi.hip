// C-ABI entry points: context, memory plumbing, Domain seam (see include/zkt_plonk.h).
#include "ctx.hpp"

#include <cstdlib>
#include "hostinv.hpp"
#include "ec.hpp"
#include "hostec.hpp"

#include <cstring>

namespace zkt {

void circuit_release(zkt_ctx* c);   // prover.hip
void msm_release(zkt_ctx* c);       // msm.hip
// msm.hip / prover.hip: a state of its own over the parent's read-only tables (zkt_ctx_fork)
int msm_fork(zkt_ctx* child, const zkt_ctx* parent);
int circuit_fork(zkt_ctx* child, const zkt_ctx* parent);

const char* exp_env(const char* name) {
#ifdef ZKT_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

int set_err(zkt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int refuse_if_forked(zkt_ctx* c, const char* what) {
    if (c && c->forks.load() > 0)
        return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string(what) + ": contexts forked from this one still use its tables (destroy them first)");
    return ZKT_OK;
}
int hip_fail(zkt_ctx* c, hipError_t e, const char* what) {
    return set_err(c, ZKT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int dev_alloc(zkt_ctx* c, void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    ZKT_HIP(c, hipMalloc(p, bytes));
    c->owned.push_back(*p);
    return ZKT_OK;
}
void dev_free(zkt_ctx* c, void* p) {
    if (!p) return;
    for (size_t i = 0; i < c->owned.size(); ++i)
        if (c->owned[i] == p) {
            c->owned[i] = c->owned.back();
            c->owned.pop_back();
            (void)hipFree(p);
            return;
        }
}
int ensure_buffer(zkt_ctx* c, void** p, size_t* cur, size_t bytes) {
    if (*cur >= bytes && *p) return ZKT_OK;
    if (*p) {
        ZKT_HIP(c, hipStreamSynchronize(c->stream));
        dev_free(c, *p);
        *p = nullptr;
        *cur = 0;
    }
    int rc = dev_alloc(c, p, bytes);
    if (rc) return rc;
    *cur = bytes;
    return ZKT_OK;
}

int comm_all_gather_host(zkt_ctx* c, const void* send, void* recv, size_t bytes) {
    if (!c->sharded() || !c->comm.vt.all_gather) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "no communicator attached");
    ++c->comm.calls;
    c->comm.bytes += bytes;
    if (c->comm.vt.all_gather(c->comm.vt.user, send, recv, bytes, 0, nullptr))
        return set_err(c, ZKT_ERR_COMM, "communicator: all_gather (host) failed");
    return ZKT_OK;
}
int comm_all_gather_dev(zkt_ctx* c, const void* d_send, void* d_recv, size_t bytes) {
    if (!c->sharded() || !c->comm.vt.all_gather) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "no communicator attached");
    const int world = c->comm.vt.world;
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    ++c->comm.calls;
    c->comm.bytes += bytes;
    if (c->comm.vt.device_buffers) {
        if (c->comm.vt.all_gather(c->comm.vt.user, d_send, d_recv, bytes, 1, (void*)c->stream))
            return set_err(c, ZKT_ERR_COMM, "communicator: all_gather (device) failed");
        return ZKT_OK;
    }
    // host-only transport (gloo rehearsals, a foreign host): stage through pinned memory
    const size_t need = bytes * (size_t)(world + 1);
    if (c->comm.pinned_bytes < need) {
        if (c->comm.pinned) (void)hipHostFree(c->comm.pinned);
        c->comm.pinned = nullptr;
        c->comm.pinned_bytes = 0;
        ZKT_HIP(c, hipHostMalloc(&c->comm.pinned, need));
        c->comm.pinned_bytes = need;
    }
    char* hs = (char*)c->comm.pinned;
    char* hr = hs + bytes;
    ZKT_HIP(c, hipMemcpyAsync(hs, d_send, bytes, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->comm.vt.all_gather(c->comm.vt.user, hs, hr, bytes, 0, nullptr))
        return set_err(c, ZKT_ERR_COMM, "communicator: all_gather (staged) failed");
    ZKT_HIP(c, hipMemcpyAsync(d_recv, hr, bytes * world, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

bool comm_async_available(const zkt_ctx* c) {
    return c->sharded() && c->comm.vt.device_buffers && c->comm.vt.all_gather_async != nullptr;
}
// stream-ordered exchange of device data through the communicator's optional asynchronous entry: no host wait on either side
int comm_all_gather_async(zkt_ctx* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t st) {
    if (!comm_async_available(c)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "communicator without all_gather_async");
    ++c->comm.calls;
    c->comm.bytes += bytes;
    if (c->comm.vt.all_gather_async(c->comm.vt.user, d_send, d_recv, bytes, (void*)st))
        return set_err(c, ZKT_ERR_COMM, "communicator: all_gather_async failed");
    return ZKT_OK;
}

static hipEvent_t prof_event(zkt_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(zkt_ctx* ctx, const char* name, hipStream_t on_stream, uint64_t units) : c(ctx), count(units) {
    if (!c->prof_on) return;
    // every event pair is a few microseconds of stream time: the light level keeps only the dominant kernel's scope, so
    // that a throughput measurement can time that kernel live without paying for ~80 other records per proof
    if (c->prof_on == 2 && strcmp(name, "msm_accumulate") != 0 && strcmp(name, "host_wait") != 0) return;
    stream = on_stream ? on_stream : c->stream;
    slot = &c->prof[name];
    e0 = prof_event(c);
    e1 = prof_event(c);
    (void)hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
    if (!slot) return;
    (void)hipEventRecord(e1, stream);
    slot->pending.emplace_back(e0, e1);
    slot->calls += count;
    ++slot->launches;
}
static void prof_resolve(zkt_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();  // scopes may have been recorded on the MSM side stream
    for (auto& kv : c->prof) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.total_ms += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

template <class P>
__global__ void k_fr_mul(const Fe<P>* a, const Fe<P>* b, Fe<P>* o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store<P>(o + i, fe_mul<P>(fe_load<P>(a + i), fe_load<P>(b + i)));
}

template <class P>
static int dump_params(uint32_t* out, size_t words) {
    const int N = P::N;
    if (words < (size_t)(3 * N + 1)) return -1;
    for (int i = 0; i < N; ++i) out[i] = P::mod(i);
    out[N] = P::INV;
    for (int i = 0; i < N; ++i) out[N + 1 + i] = P::one(i);
    for (int i = 0; i < N; ++i) out[2 * N + 1 + i] = P::r2(i);
    return N;
}

}  // namespace zkt

using namespace zkt;

extern "C" {

const char* zkt_version(void) { return "zkt-plonk_amd 0.1 (gfx950)"; }

int zkt_ctx_create(int curve_id, int device_id, zkt_ctx** out) {
    if (!out) return ZKT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (curve_id != ZKT_CURVE_BN254 && curve_id != ZKT_CURVE_BLS12_381) return ZKT_ERR_INVALID_ARGUMENT;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count)
        return ZKT_ERR_NO_DEVICE;  // no CPU fallback by design
    if (hipSetDevice(device_id) != hipSuccess) return ZKT_ERR_NO_DEVICE;
    zkt_ctx* c = new zkt_ctx();
    c->batch_off = exp_env("ZKT_MSM_NO_BATCH") != nullptr;
    c->aux_off = exp_env("ZKT_NO_AUX") != nullptr;
    c->curve = curve_id;
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ZKT_ERR_HIP;
    }
    c->own_stream = true;
    *out = c;
    return ZKT_OK;
}

int zkt_ctx_fork(zkt_ctx* parent, zkt_ctx** out) {
    if (!parent || !out) return ZKT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (parent->sharded()) return set_err(parent, ZKT_ERR_INVALID_ARGUMENT, "zkt_ctx_fork: not for a context with a communicator");
    if (parent->zombie) return set_err(parent, ZKT_ERR_INVALID_ARGUMENT, "zkt_ctx_fork: the context has been destroyed");
    zkt_ctx* root = parent->parent ? parent->parent : parent;   // forks of a fork hang off the owner of the tables
    zkt_ctx* c = nullptr;
    int rc = zkt_ctx_create(parent->curve, parent->device, &c);
    if (rc) return rc;
    (void)hipStreamSynchronize(parent->stream);   // the tables are complete
    c->lagrange_off = parent->lagrange_off;
    c->batch_off = parent->batch_off;
    c->aux_off = parent->aux_off;
    c->ntt_plans = parent->ntt_plans;             // twiddle tables: immutable, owned by the root
    c->parent = root;
    root->forks.fetch_add(1);
    if ((rc = msm_fork(c, parent)) || (rc = circuit_fork(c, parent))) {
        parent->err = c->err;
        zkt_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return ZKT_OK;
}

void zkt_ctx_destroy(zkt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->forks.load() > 0) {   // forks still read this context's tables: linger until the last of them is gone
        c->zombie = true;
        return;
    }
    zkt_ctx* const owner = c->parent;
    circuit_release(c);
    msm_release(c);
    prof_resolve(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->event_pool.clear();
    c->ntt_plans.clear();
    c->msm.reset();
    c->circuit.reset();
    if (c->comm.pinned) (void)hipHostFree(c->comm.pinned);
    for (void* p : c->owned) (void)hipFree(p);
    c->owned.clear();
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    if (owner && owner->forks.fetch_sub(1) == 1 && owner->zombie) {
        owner->zombie = false;
        zkt_ctx_destroy(owner);
    }
}

int zkt_ctx_set_comm(zkt_ctx* c, const zkt_comm_vtable* comm) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    if (int rf = refuse_if_forked(c, "zkt_ctx_set_comm")) return rf;
    if (c->parent && comm && comm->world > 1) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "a forked context cannot take a communicator");
    if (comm && comm->world > 1) {
        const int w = comm->world;
        if ((w != 2 && w != 4 && w != 8) || comm->rank < 0 || comm->rank >= w || !comm->all_gather)
            return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "communicator: world must be 1, 2, 4 or 8 with a valid rank and callback");
    }
    // keys are laid out for the rank's share: whatever was loaded for another layout goes
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    circuit_release(c);
    msm_release(c);
    c->comm.vt = zkt_comm_vtable{};
    c->comm.vt.world = 1;
    if (comm && comm->world > 1) c->comm.vt = *comm;
    c->comm.calls = 0;
    c->comm.bytes = 0;
    return ZKT_OK;
}
int zkt_shard_range(size_t total, int rank, int world, size_t* lo, size_t* hi) {
    if (world < 1 || rank < 0 || rank >= world || !lo || !hi) return ZKT_ERR_INVALID_ARGUMENT;
    const size_t base = total / (size_t)world, rem = total % (size_t)world;
    *lo = (size_t)rank * base + ((size_t)rank < rem ? (size_t)rank : rem);
    *hi = *lo + base + ((size_t)rank < rem ? 1 : 0);
    return ZKT_OK;
}
int zkt_comm_stats(zkt_ctx* c, uint64_t* calls, uint64_t* bytes_sent) {
    if (!c || !calls || !bytes_sent) return ZKT_ERR_INVALID_ARGUMENT;
    *calls = c->comm.calls;
    *bytes_sent = c->comm.bytes;
    return ZKT_OK;
}
int zkt_comm_selftest(const zkt_comm_vtable* comm, const void* send, void* recv, size_t bytes) {
    if (!comm || !comm->all_gather || !send || !recv) return ZKT_ERR_INVALID_ARGUMENT;
    return comm->all_gather(comm->user, send, recv, bytes, 0, nullptr) ? ZKT_ERR_COMM : ZKT_OK;
}

const char* zkt_last_error(const zkt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int zkt_ctx_set_stream(zkt_ctx* c, void* hip_stream) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return ZKT_OK;
}

int zkt_profile_enable(zkt_ctx* c, int on) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    prof_resolve(c);
    if (on) c->prof.clear();
    c->prof_on = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return ZKT_OK;
}
int zkt_profile_get(zkt_ctx* c, const char* name, uint64_t* calls, double* total_ms) {
    if (!c || !name || !calls || !total_ms) return ZKT_ERR_INVALID_ARGUMENT;
    prof_resolve(c);
    // "<scope>#launches": the number of scopes recorded instead of the units they stand for (a batched launch of three
    // MSMs is one scope of three units)
    std::string key(name);
    const size_t hash = key.find("#launches");
    const bool want_launches = hash != std::string::npos;
    if (want_launches) key.resize(hash);
    auto it = c->prof.find(key);
    if (it == c->prof.end()) {
        *calls = 0;
        *total_ms = 0.0;
        return ZKT_OK;
    }
    *calls = want_launches ? it->second.launches : it->second.calls;
    *total_ms = it->second.total_ms;
    return ZKT_OK;
}

int zkt_ctx_synchronize(zkt_ctx* c) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

int zkt_dev_alloc(zkt_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return ZKT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    return dev_alloc(c, dptr, bytes);
}
int zkt_dev_free(zkt_ctx* c, void* dptr) {
    if (!c) return ZKT_ERR_INVALID_ARGUMENT;
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    dev_free(c, dptr);
    return ZKT_OK;
}
int zkt_dev_upload(zkt_ctx* c, void* dptr, const void* host, size_t bytes) {
    if (!c || (!dptr && bytes) || (!host && bytes)) return ZKT_ERR_INVALID_ARGUMENT;
    ZKT_HIP(c, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}
int zkt_dev_download(zkt_ctx* c, void* host, const void* dptr, size_t bytes) {
    if (!c || (!dptr && bytes) || (!host && bytes)) return ZKT_ERR_INVALID_ARGUMENT;
    ZKT_HIP(c, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

int zkt_ntt_dev(zkt_ctx* c, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out) {
    if (!c || !d_out || (!d_in && in_len)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    (void)hipSetDevice(c->device);
    return ntt_run(c, log_n, inverse, coset ? 1 : 0, d_in, in_len, d_out);
}

int zkt_ntt(zkt_ctx* c, int log_n, int inverse, int coset, const uint64_t* in, size_t in_len, uint64_t* out) {
    if (!c || !out || (!in && in_len)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (log_n < 0 || log_n > 27) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "InvalidEvalDomainSize");
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)1 << log_n;
    if (in_len > n) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "more coefficients than the domain size");
    int rc = ensure_buffer(c, &c->io_a, &c->io_a_bytes, n * 32);
    if (rc) return rc;
    if (in_len) ZKT_HIP(c, hipMemcpyAsync(c->io_a, in, in_len * 32, hipMemcpyHostToDevice, c->stream));
    rc = ntt_run(c, log_n, inverse, coset ? 1 : 0, c->io_a, in_len, c->io_a);
    if (rc) return rc;
    ZKT_HIP(c, hipMemcpyAsync(out, c->io_a, n * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

int zkt_ntt_class(zkt_ctx* c, int log_n, int log_big, int cls, const uint64_t* in, size_t in_len, uint64_t* out) {
    if (!c || !out || (!in && in_len)) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    if (log_n < 0 || log_n > 27) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "InvalidEvalDomainSize");
    (void)hipSetDevice(c->device);
    const size_t n = (size_t)1 << log_n;
    int rc = ensure_buffer(c, &c->io_a, &c->io_a_bytes, (in_len > n ? in_len : n) * 32);
    if (rc) return rc;
    if ((rc = ensure_buffer(c, &c->io_b, &c->io_b_bytes, n * 32))) return rc;
    if (in_len) ZKT_HIP(c, hipMemcpyAsync(c->io_a, in, in_len * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = ntt_run_class(c, log_n, log_big, cls, c->io_a, in_len, c->io_a, c->io_b))) return rc;
    ZKT_HIP(c, hipMemcpyAsync(out, c->io_a, n * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

int zkt_domain_group_gen(zkt_ctx* c, int log_n, uint64_t* out4) {
    if (!c || !out4) return ZKT_ERR_INVALID_ARGUMENT;
    if (c->curve == ZKT_CURVE_BN254) {
        if (log_n < 0 || log_n > Bn254Fr::TWO_ADICITY) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "InvalidEvalDomainSize");
        Fe<Bn254Fr> w = root_of_unity<Bn254Fr>(log_n);
        memcpy(out4, w.v, 32);
    } else {
        if (log_n < 0 || log_n > Bls381Fr::TWO_ADICITY) return set_err(c, ZKT_ERR_INVALID_DOMAIN_SIZE, "InvalidEvalDomainSize");
        Fe<Bls381Fr> w = root_of_unity<Bls381Fr>(log_n);
        memcpy(out4, w.v, 32);
    }
    return ZKT_OK;
}

int zkt_debug_params(zkt_ctx* c, int which, uint32_t* out, size_t out_words) {
    if (!c || !out) return -1;
    if (c->curve == ZKT_CURVE_BN254) return which == 0 ? dump_params<Bn254Fr>(out, out_words) : dump_params<Bn254Fq>(out, out_words);
    return which == 0 ? dump_params<Bls381Fr>(out, out_words) : dump_params<Bls381Fq>(out, out_words);
}

}  // extern "C"

// Host-side execution of the field routines (same __host__ __device__ code the kernels run): lets the
// CPU test-suite pin the 29-bit-limb arithmetic against big integers without a GPU.
//   op 0: packed Montgomery product (fe_mul)          op 1: 32-bit-limb CIOS reference (fe_mul_sat)
//   op 2: ark -> R'-limbs -> ark round trip           op 3: lazy chain  (a + b) * (a + 8p - b) reduced
//   op 4: a^-1 by the host's binary GCD (hostinv.hpp)  op 5: a^-1 by the kernels' Fermat ladder
//   op 6: double product a*a + b*b with one reduction (fx_mul2_inl) and the squaring kernel, lazily summed
//   op 7: (a + 3p - b) * b through the carry-free difference (fx_sub_lazy)
template <class P>
static void host_field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    Fe<P> x, y, r;
    memcpy(x.v, a, P::N * 4);
    memcpy(y.v, b, P::N * 4);
    if (op == 0) {
        r = fe_mul<P>(x, y);
    } else if (op == 1) {
        r = fe_mul_sat<P>(x, y);
    } else if (op == 2) {
        r = fx_to_ark<P>(fx_from_ark<P>(x));
    } else if (op == 4) {
        r = fe_inv_host<P>(x);
    } else if (op == 5) {
        r = fe_inv<P>(x);
    } else if (op == 6) {
        const Fx<P> xa = fx_from_ark<P>(x), ya = fx_from_ark<P>(y);
        const Fx<P> two = fx_mul2_inl<P>(xa, xa, ya, ya);                        // a^2 + b^2
        const Fx<P> sq = fx_add<P>(fx_sqr_inl<P>(xa), fx_sqr_inl<P>(ya));       // the same, two reductions
        r = fx_to_ark<P>(fx_sub<P, 4>(fx_add<P>(two, two), sq));               // 2 (a^2+b^2) - (a^2+b^2)
    } else if (op == 8) {
        // the NTT butterfly's arithmetic: lazy sums, the wide carry-free difference, the product by a constant with a
        // precomputed quotient.  2 (a - b) * y, y the plain value of b
        const Fx<P> w = fx_unpack<P>(fe_from_mont<P>(y));
        const Fx<P> wq = fx_mul_low<P>(fx_cond_sub_p<P>(fx_from_ark<P>(y)), fx_neg_p_inverse<P>());
        const Fx<P> u = fx_unpack<P>(x), v = fx_unpack<P>(y);
        const Fx<P> d = fx_sub_lazy_wide<P, 8, 30>(fx_add_lazy<P>(u, u), fx_add_lazy<P>(v, v));
        r = fx_pack<P>(fx_cond_sub_p<P>(fx_cond_sub_p<P>(fx_mul_shoup<P>(d, w, wq))));
    } else if (op == 9) {
        // 4 (a + b) through two levels of limb-wise sums and the lazy reduction (fields whose top limb is wide enough for
        // its quotient estimate: all but the 381-bit one)
        if constexpr (FxP<P>::mod(FxP<P>::L - 1) >= (1u << 16)) {
            const Fx<P> u = fx_unpack<P>(x), v = fx_unpack<P>(y);
            const Fx<P> s = fx_add_lazy<P>(u, v);                       // < 2p, limbs < 2^30
            const Fx<P> t = fx_reduce_lazy<P>(fx_add_lazy<P>(s, s));    // 2 (a + b), limbs < 2^31 -> < 3p, normalised
            r = fx_pack<P>(fx_cond_sub_p<P>(fx_cond_sub_p<P>(fx_reduce_lazy<P>(fx_add_lazy<P>(t, t)))));
        } else {
            r = fe_dbl<P>(fe_dbl<P>(fe_add<P>(x, y)));
        }
    } else if (op == 7) {
        const Fx<P> xa = fx_from_ark<P>(x), ya = fx_cond_sub_p<P>(fx_from_ark<P>(y));
        r = fx_to_ark<P>(fx_mul<P>(ya, fx_sub_lazy<P, 3>(xa, ya)));             // (a - b) b
    } else {
        Fx<P> xa = fx_from_ark<P>(x), ya = fx_from_ark<P>(y);
        Fx<P> s = fx_add<P>(xa, ya);              // < 4p
        Fx<P> d = fx_sub<P, 8>(xa, ya);           // < 10p
        Fx<P> m = fx_mul<P>(s, d);                // (a+b)(a-b) in R' form, < 2p
        r = fx_to_ark<P>(fx_add<P>(fx_add<P>(m, m), m));  // 3 (a^2 - b^2), lazily < 6p
    }
    memcpy(out, r.v, P::N * 4);
}

template <class C>
static void g1_sum_host(const uint64_t* pts, size_t count, uint64_t* out, int* out_inf) {
    using Q = typename C::Fq;
    constexpr int L = Q::N / 2;   // u64 limbs per coordinate
    // the host's own 64-bit-limb arithmetic (hostec.hpp), the one that finishes every MSM's bucket reduction
    hostec::HX<Q> acc = hostec::hx_identity<Q>();
    for (size_t i = 0; i < count; ++i) {
        Affine<Q> a;
        memcpy(a.x.v, pts + i * 2 * L, Q::N * 4);
        memcpy(a.y.v, pts + i * 2 * L + L, Q::N * 4);
        if (aff_is_inf<Q>(a)) continue;
        acc = hostec::hx_add<Q>(acc, hostec::hx_from<Q>(xyzz_from_affine<Q>(a)));
    }
    const Affine<Q> r = xyzz_to_affine_host<Q>(hostec::hx_to<Q>(acc));
    memcpy(out, r.x.v, Q::N * 4);
    memcpy(out + L, r.y.v, Q::N * 4);
    if (out_inf) *out_inf = aff_is_inf<Q>(r) ? 1 : 0;
}

extern "C" {

int zkt_g1_sum_host(int curve_id, const uint64_t* pts, size_t count, uint64_t* out, int* out_inf) {
    if ((!pts && count) || !out) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) g1_sum_host<Bn254Curve>(pts, count, out, out_inf);
    else if (curve_id == ZKT_CURVE_BLS12_381) g1_sum_host<Bls381Curve>(pts, count, out, out_inf);
    else return ZKT_ERR_INVALID_ARGUMENT;
    return ZKT_OK;
}

int zkt_host_field_op(int curve_id, int which, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    if (!a || !b || !out) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) {
        if (which == 0) host_field_op<Bn254Fr>(op, a, b, out); else host_field_op<Bn254Fq>(op, a, b, out);
    } else {
        if (which == 0) host_field_op<Bls381Fr>(op, a, b, out); else host_field_op<Bls381Fq>(op, a, b, out);
    }
    return ZKT_OK;
}

int zkt_debug_fr_mul(zkt_ctx* c, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    if (!c || !a || !b || !out) return ZKT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    int rc = ensure_buffer(c, &c->io_a, &c->io_a_bytes, n * 32);
    if (rc) return rc;
    rc = ensure_buffer(c, &c->io_b, &c->io_b_bytes, n * 32);
    if (rc) return rc;
    ZKT_HIP(c, hipMemcpyAsync(c->io_a, a, n * 32, hipMemcpyHostToDevice, c->stream));
    ZKT_HIP(c, hipMemcpyAsync(c->io_b, b, n * 32, hipMemcpyHostToDevice, c->stream));
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (c->curve == ZKT_CURVE_BN254)
        hipLaunchKernelGGL(k_fr_mul<Bn254Fr>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<Bn254Fr>*)c->io_a,
                           (const Fe<Bn254Fr>*)c->io_b, (Fe<Bn254Fr>*)c->io_a, n);
    else
        hipLaunchKernelGGL(k_fr_mul<Bls381Fr>, dim3(blocks), dim3(256), 0, c->stream, (const Fe<Bls381Fr>*)c->io_a,
                           (const Fe<Bls381Fr>*)c->io_b, (Fe<Bls381Fr>*)c->io_a, n);
    ZKT_HIP(c, hipGetLastError());
    ZKT_HIP(c, hipMemcpyAsync(out, c->io_a, n * 32, hipMemcpyDeviceToHost, c->stream));
    ZKT_HIP(c, hipStreamSynchronize(c->stream));
    return ZKT_OK;
}

}  // extern "C"
