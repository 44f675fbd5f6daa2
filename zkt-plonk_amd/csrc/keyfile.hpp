// Sequential reader of the reference CLI's --epk file (keyfile.hip; bin/src/main.rs:34-35,108-109): plonk-core's
// ExtendedProverKey<F> (keys/mod.rs:148-174) is seventeen Vec<F> in declaration order, each a u64 length and the canonical
// little-endian values.  Host only.
#pragma once
#include <cstdint>
#include <cstdio>

namespace zkt {

// arith { q_m_coset q_l_coset q_r_coset q_o_coset q_c_coset } (keys/arithmetic.rs:51-62), lookup { q_lookup q_lookup_coset
// q_table_coset } (keys/lookup.rs:70-77), perm { sigma1 sigma1_coset sigma2 sigma2_coset sigma3 sigma3_coset x_coset }
// (keys/permutation.rs:74-92), zh_coset, l_1_coset
constexpr int EPK_VECTORS = 17;
enum { EPK_QM_C = 0, EPK_QL_C, EPK_QR_C, EPK_QO_C, EPK_QC_C, EPK_QLOOKUP, EPK_QLOOKUP_C, EPK_QTABLE_C, EPK_S1, EPK_S1_C, EPK_S2,
       EPK_S2_C, EPK_S3, EPK_S3_C, EPK_X_C, EPK_ZH_C, EPK_L1_C };

struct EpkReader {
    FILE* f = nullptr;
    uint64_t size = 0, pos = 0, left = 0;   // left: elements of the current vector not yet read
    bool open(const char* path);
    void close();
    bool next(uint64_t* len);                  // the next vector's length (the previous one must be used up)
    bool read(uint8_t* dst, size_t elems);     // canonical bytes of the next `elems` elements, 32 each
    bool skip();                               // the rest of the current vector
    bool at_end() const { return left == 0 && pos == size; }
};

}  // namespace zkt
