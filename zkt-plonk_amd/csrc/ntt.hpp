// Radix-2 NTT over the scalar field, natural-order in and out, for gfx950.
//
// Replaces ark-poly 0.3 Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place as
// reached from plonk-core/src/util.rs:63-140 (a1-a4 of SURVEY.md section 8).
//
// Decomposition.  N = R1*R2(*R3), 32 <= Ri <= 512.  Input index n = n1*S1 + n2*S2 + n3,
// output index k = k1 + R1*k2 + R1*R2*k3.  Pass i transforms the Ri-point axis of a
// [Ri][T] tile that lives in LDS (T consecutive columns, Ri*T = 1024 elements = 32 KiB), reading
// and writing runs of T*32 B.  Between passes the element is multiplied by
// w_{P(i+1)}^{n(i+1)*K(i)} (K(i) = partial output index): the next pass stages that row of
// twiddles (Ri values shared by all T columns of the tile) straight from an L2-resident table;
// only the last pass, whose tile spans T different K, reads a full-size table.  Coset shifts,
// the 1/N of the inverse and the g^-k of the inverse coset transform are folded into those tables
// (plus one Ri-entry row table), so they cost no extra pass over HBM.
// The last pass writes the digit-reversed (= natural) order directly: [k1][k2][k3] -> k1+R1*k2+R1*R2*k3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace zkt {

constexpr int NTT_MAX_BATCH = 4;   // transforms of one plan issued as ONE launch per pass (gridDim.y = polynomial)

struct NttPassArgs {
    const void* in[NTT_MAX_BATCH];    // Fe* (raw passes: polynomial y lives at in[0] + y * 36 * raw_n bytes)
    void* out[NTT_MAX_BATCH];         // Fe*
    const void* w_inner;   // R/2 inner twiddles W_R^j, Montgomery form (packed words)
    const void* w_inner_s; // the same as (w, floor(w 2^261 / p)) pairs of 9 + 9 limbs (fx_mul_shoup)
    const void* in_row;    // optional R-entry input row scale (forward coset, pass 1)
    const void* tw;        // optional boundary twiddles (row-shared or tile-shaped)
    const void* out_row;   // optional R-entry output row scale (inverse coset, last pass)
    uint64_t in_len[NTT_MAX_BATCH];   // elements of `in` that exist (rest read as zero); pass 1 only
    uint32_t log_n;
    uint32_t log_s;        // non-last: log2 of the inner stride S (columns); last: unused
    uint32_t log_r1;       // last: log2 R1
    uint32_t log_mid;      // last: log2 (N / (R1*Rp))
    // Between passes the data stay as 29-bit limbs, lazily reduced (< 3p), in three planes of N entries (limbs 0-3,
    // limbs 4-7, limb 8: the LDS tile's layout, 36 N bytes at raw_base): no packing, reduction or unpacking there.
    uint32_t in_raw, out_raw;
    uint64_t raw_n;        // N (plane stride)
};

template <class P>
struct NttPlan {
    int log_n = -1;
    int inverse = 0, coset = 0;
    int npass = 0;            // 0: single-workgroup kernel (log_n <= 10)
    int log_r[3] = {0, 0, 0};
    void* w_inner[3] = {nullptr, nullptr, nullptr};
    void* w_inner_s[3] = {nullptr, nullptr, nullptr};
    void* in_row = nullptr;   // pass 1 (forward coset)
    void* tw[3] = {nullptr, nullptr, nullptr};  // tw[i] consumed by pass i (i >= 1)
    void* out_row = nullptr;  // last pass (inverse coset)
    // single-workgroup path
    void* small_w = nullptr;    // N/2 powers of the root
    void* small_in = nullptr;   // optional N-entry input scale
    void* small_out = nullptr;  // optional N-entry output scale
    size_t table_bytes = 0;
};

}  // namespace zkt
