// Library context: one per caller thread / proof stream (SURVEY.md section 8b "Threading").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include <memory>
#include <atomic>

#include "../../include/zkt_plonk.h"
#include "fp.hpp"
#include "fx.hpp"
#include "ntt.hpp"

namespace zkt {

struct MsmState;      // msm.hip
struct CircuitState;  // prover.hip

}  // namespace zkt

struct zkt_ctx {
    int curve = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // NTT plans keyed by (log_n, inverse, coset); tables live in HBM for the ctx lifetime.
    std::map<std::tuple<int, int, int>, std::shared_ptr<void>> ntt_plans;
    // scratch (grown on demand, never shrunk)
    void* ntt_scratch = nullptr;
    size_t ntt_scratch_bytes = 0;
    void* io_a = nullptr;
    size_t io_a_bytes = 0;
    void* io_b = nullptr;
    size_t io_b_bytes = 0;

    // optional per-kernel HIP-event timing (bench.py's live roofline measurement)
    int prof_on = 0;   // 0 off, 1 every scope, 2 only the dominant kernel's scope ("msm_accumulate")
    struct ProfSlot {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
        double total_ms = 0.0;
        uint64_t calls = 0;      // units (a batched launch of three MSMs counts three)
        uint64_t launches = 0;   // scopes recorded (zkt_profile_get "<name>#launches")
    };
    std::map<std::string, ProfSlot> prof;
    std::vector<hipEvent_t> event_pool;

    // Communicator of a proof sharded across GPUs (zkt_ctx_set_comm); world == 1: single GPU
    struct Comm {
        zkt_comm_vtable vt{};
        void* pinned = nullptr;      // staging of device exchanges when the communicator takes host buffers only
        size_t pinned_bytes = 0;
        uint64_t calls = 0, bytes = 0;
    } comm;
    bool sharded() const { return comm.vt.world > 1; }

    std::shared_ptr<zkt::MsmState> msm;
    uint64_t msm_epoch = 0;   // bumped by every MSM enqueue and SRS (re)load: work issued ahead of time is tied to it
    uint64_t srs_generation = 0;   // bumped by every SRS (re)load: cached commitments are tied to the key they were made under
    // zkt_ctx_fork: forks share this context's read-only tables.  A parent with live forks refuses to reload them; when it
    // is destroyed first it lingers (zombie) until its last fork is gone.
    zkt_ctx* parent = nullptr;
    std::atomic<int> forks{0};
    bool zombie = false;
    bool aux_off = false;          // A/B builds: ZKT_NO_AUX keeps round 5's second opening on the main stream
    bool batch_off = false;        // A/B builds: ZKT_MSM_NO_BATCH commits a round's polynomials one launch sequence each
    bool lagrange_off = false;     // zkt_ctx_set_lagrange(ctx, 0): evaluations are committed through their coefficients
    std::shared_ptr<zkt::CircuitState> circuit;
    std::vector<void*> owned;  // every hipMalloc made on behalf of this ctx
};

namespace zkt {

// Experiment knobs (A/B runs of tools/ab_*.sh) are read from the environment ONLY by libraries built with
// -DZKT_EXPERIMENTS (zkt-plonk_amd/build.py build_experiments() -> _ab/libzkt_exp.so); the shipped library never looks
// at the environment, so a stray variable cannot change or break a proof.
const char* exp_env(const char* name);

int set_err(zkt_ctx* c, int code, const std::string& msg);
// ZKT_ERR_INVALID_ARGUMENT when forks of `c` are alive (their tables are c's)
int refuse_if_forked(zkt_ctx* c, const char* what);
int hip_fail(zkt_ctx* c, hipError_t e, const char* what);

#define ZKT_HIP(c, call)                                        \
    do {                                                        \
        hipError_t _e = (call);                                 \
        if (_e != hipSuccess) return zkt::hip_fail((c), _e, #call); \
    } while (0)

// Records a start/stop HIP event pair on the context's stream around a launch (or a group of
// launches) when profiling is on; durations are resolved lazily by zkt_profile_get.
struct ProfScope {
    zkt_ctx* c;
    zkt_ctx::ProfSlot* slot = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream = nullptr;
    uint64_t count = 1;   // units the scope stands for (a batch of transforms is counted per transform)
    ProfScope(zkt_ctx* ctx, const char* name, hipStream_t on_stream = nullptr, uint64_t units = 1);
    ~ProfScope();
};

// all-gather through the context's communicator: `bytes` per rank, rank order.  _host: host buffers;
// _dev: device buffers, ordered after everything enqueued on the context's stream, complete on return
int comm_all_gather_host(zkt_ctx* c, const void* send, void* recv, size_t bytes);
int comm_all_gather_dev(zkt_ctx* c, const void* d_send, void* d_recv, size_t bytes);
// the communicator's optional stream-ordered entry (zkt_comm_vtable::all_gather_async): enqueued on `st`, no host wait
bool comm_async_available(const zkt_ctx* c);
int comm_all_gather_async(zkt_ctx* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t st);

// grows *p to at least `bytes`
int ensure_buffer(zkt_ctx* c, void** p, size_t* cur, size_t bytes);
int dev_alloc(zkt_ctx* c, void** p, size_t bytes);
void dev_free(zkt_ctx* c, void* p);

// ntt.hip
template <class P>
Fe<P> root_of_unity(int log_n);
// coset: 0 none, 1 the generator g, ntt_class_code(log_big, cls) a class of a larger coset (see ntt.hip)
int ntt_run(zkt_ctx* c, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out);
// nb <= NTT_MAX_BATCH transforms of one plan, one launch per pass; distinct outputs, out[y] == in[y] allowed
int ntt_run_batch(zkt_ctx* c, int log_n, int inverse, int coset, int nb, const void* const* d_in, const size_t* in_len,
                  void* const* d_out);
int ntt_class_code(int log_big, int cls);
int ntt_run_class(zkt_ctx* c, int log_n, int log_big, int cls, const void* d_in, size_t in_len, void* d_out, void* d_fold);

}  // namespace zkt
