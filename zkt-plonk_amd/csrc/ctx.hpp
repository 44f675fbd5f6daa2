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
    bool prof_on = false;
    struct ProfSlot {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
        double total_ms = 0.0;
        uint64_t calls = 0;
    };
    std::map<std::string, ProfSlot> prof;
    std::vector<hipEvent_t> event_pool;

    std::shared_ptr<zkt::MsmState> msm;
    uint64_t msm_epoch = 0;   // bumped by every MSM enqueue and SRS (re)load: work issued ahead of time is tied to it
    uint64_t srs_generation = 0;   // bumped by every SRS (re)load: cached commitments are tied to the key they were made under
    std::shared_ptr<zkt::CircuitState> circuit;
    std::vector<void*> owned;  // every hipMalloc made on behalf of this ctx
};

namespace zkt {

int set_err(zkt_ctx* c, int code, const std::string& msg);
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
    ProfScope(zkt_ctx* ctx, const char* name, hipStream_t on_stream = nullptr);
    ~ProfScope();
};

// grows *p to at least `bytes`
int ensure_buffer(zkt_ctx* c, void** p, size_t* cur, size_t bytes);
int dev_alloc(zkt_ctx* c, void** p, size_t bytes);
void dev_free(zkt_ctx* c, void* p);

// ntt.hip
template <class P>
Fe<P> root_of_unity(int log_n);
int ntt_run(zkt_ctx* c, int log_n, int inverse, int coset, const void* d_in, size_t in_len, void* d_out);

}  // namespace zkt
