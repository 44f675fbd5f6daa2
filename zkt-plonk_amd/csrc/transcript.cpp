// Host-side transcripts: Keccak-f[1600], STROBE-128 / Merlin, Keccak-256 Ethereum transcript.
// See transcript.hpp for the reference files each class mirrors.
#include "transcript.hpp"

#include <cstring>

namespace zkt {

static inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void keccak_f1600(uint8_t state[200]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rho offsets and pi permutation in the usual "walk the (x, y) -> (y, 2x + 3y)" order
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    uint64_t a[25];
    for (int i = 0; i < 25; ++i) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; --b) v = (v << 8) | state[8 * i + b];
        a[i] = v;
    }
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) {
            uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
            for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
        }
        uint64_t cur = a[1];
        for (int i = 0; i < 24; ++i) {
            int j = PIL[i];
            uint64_t tmp = a[j];
            a[j] = rotl64(cur, ROT[i]);
            cur = tmp;
        }
        for (int y = 0; y < 25; y += 5) {
            uint64_t r0 = a[y], r1 = a[y + 1], r2 = a[y + 2], r3 = a[y + 3], r4 = a[y + 4];
            a[y] = r0 ^ (~r1 & r2);
            a[y + 1] = r1 ^ (~r2 & r3);
            a[y + 2] = r2 ^ (~r3 & r4);
            a[y + 3] = r3 ^ (~r4 & r0);
            a[y + 4] = r4 ^ (~r0 & r1);
        }
        a[0] ^= RC[round];
    }
    for (int i = 0; i < 25; ++i)
        for (int b = 0; b < 8; ++b) state[8 * i + b] = (uint8_t)(a[i] >> (8 * b));
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint8_t st[200];
    memset(st, 0, sizeof(st));
    while (len >= rate) {
        for (size_t i = 0; i < rate; ++i) st[i] ^= data[i];
        keccak_f1600(st);
        data += rate;
        len -= rate;
    }
    for (size_t i = 0; i < len; ++i) st[i] ^= data[i];
    st[len] ^= 0x01;  // legacy Keccak padding (sha3::Keccak256), not FIPS-202's 0x06
    st[rate - 1] ^= 0x80;
    keccak_f1600(st);
    memcpy(out, st, 32);
}

// ---- STROBE-128 (subset used by merlin) -----------------------------------------------------------
namespace {
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;   // FLAG_T = 8 (transport) is never used by merlin
}

Strobe128::Strobe128(const std::string& protocol_label) {
    memset(st_, 0, sizeof(st_));
    const uint8_t init[6] = {1, (uint8_t)(R + 2), 1, 0, 1, 96};
    memcpy(st_, init, 6);
    memcpy(st_ + 6, "STROBEv1.0.2", 12);
    keccak_f1600(st_);
    meta_ad(reinterpret_cast<const uint8_t*>(protocol_label.data()), protocol_label.size(), false);
}

void Strobe128::run_f() {
    st_[pos_] ^= pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[R + 1] ^= 0x80;
    keccak_f1600(st_);
    pos_ = 0;
    pos_begin_ = 0;
}

void Strobe128::absorb(const uint8_t* data, size_t len) {
    for (size_t i = 0; i < len; ++i) {
        st_[pos_] ^= data[i];
        if (++pos_ == R) run_f();
    }
}

void Strobe128::squeeze(uint8_t* out, size_t len) {
    for (size_t i = 0; i < len; ++i) {
        out[i] = st_[pos_];
        st_[pos_] = 0;
        if (++pos_ == R) run_f();
    }
}

void Strobe128::begin_op(uint8_t flags, bool more) {
    if (more) return;  // continuation of the current operation (flags must match)
    uint8_t old_begin = pos_begin_;
    pos_begin_ = (uint8_t)(pos_ + 1);
    cur_flags_ = flags;
    const uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    const bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
    if (force_f && pos_ != 0) run_f();
}

void Strobe128::meta_ad(const uint8_t* data, size_t len, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(data, len);
}
void Strobe128::ad(const uint8_t* data, size_t len, bool more) {
    begin_op(FLAG_A, more);
    absorb(data, len);
}
void Strobe128::prf(uint8_t* out, size_t len, bool more) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, more);
    squeeze(out, len);
}

// ---- merlin::Transcript --------------------------------------------------------------------------
Merlin::Merlin(const std::string& label) : strobe_("Merlin v1.0") {
    append_message("dom-sep", reinterpret_cast<const uint8_t*>(label.data()), label.size());
}
void Merlin::append_message(const std::string& label, const uint8_t* msg, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe_.meta_ad(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
    strobe_.meta_ad(l4, 4, true);
    strobe_.ad(msg, len, false);
}
void Merlin::append_u64(const std::string& label, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(x >> (8 * i));
    append_message(label, b, 8);
}
void Merlin::challenge_bytes(const std::string& label, uint8_t* out, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe_.meta_ad(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
    strobe_.meta_ad(l4, 4, true);
    strobe_.prf(out, len, false);
}

// ---- plonk-core/src/transcript.rs:46-109 ------------------------------------------------------------
void MerlinHostTranscript::append_u64(const char* label, uint64_t v) { t.append_u64(label, v); }
void MerlinHostTranscript::append_scalars(const char* label, const uint8_t* le, size_t count, size_t fr_bytes, bool) {
    t.append_message(label, le, count * fr_bytes);  // one message, scalars back to back (transcript.rs:69-79)
}
void MerlinHostTranscript::append_commitment(const char* label, const uint8_t* x_le, const uint8_t* y_le,
                                             size_t fq_bytes, bool infinity) {
    // GroupAffine::write: x || y || infinity (transcript.rs:81-86); zero() is (0, 1, true)
    std::vector<uint8_t> buf(2 * fq_bytes + 1, 0);
    if (infinity) {
        buf[fq_bytes] = 1;
        buf[2 * fq_bytes] = 1;
    } else {
        memcpy(buf.data(), x_le, fq_bytes);
        memcpy(buf.data() + fq_bytes, y_le, fq_bytes);
    }
    t.append_message(label, buf.data(), buf.size());
}
void MerlinHostTranscript::challenge_scalar(const char* label, size_t fr_bits, uint8_t out_le[32]) {
    const size_t nbytes = (fr_bits + 7) / 8 - 1;  // transcript.rs:102
    memset(out_le, 0, 32);
    t.challenge_bytes(label, out_le, nbytes);     // from_random_bytes: little-endian integer < 2^248
}

// ---- gadgets/src/transcript.rs:8-90 --------------------------------------------------------------------
EthereumHostTranscript::EthereumHostTranscript() {
    memset(state0_, 0, 32);
    memset(state1_, 0, 32);
}
void EthereumHostTranscript::append_bytes(const uint8_t* item, size_t len) {
    std::vector<uint8_t> data(1 + 64 + len);
    memcpy(data.data() + 1, state0_, 32);
    memcpy(data.data() + 33, state1_, 32);
    memcpy(data.data() + 65, item, len);
    uint8_t n0[32], n1[32];
    data[0] = 0;
    keccak256(data.data(), data.size(), n0);
    data[0] = 1;
    keccak256(data.data(), data.size(), n1);
    memcpy(state0_, n0, 32);
    memcpy(state1_, n1, 32);
}
void EthereumHostTranscript::append_u64(const char*, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * (7 - i)));
    append_bytes(b, 8);
}
void EthereumHostTranscript::append_scalars(const char*, const uint8_t* le, size_t count, size_t fr_bytes, bool) {
    for (size_t k = 0; k < count; ++k) {
        uint8_t be[64];
        for (size_t i = 0; i < fr_bytes; ++i) be[i] = le[k * fr_bytes + (fr_bytes - 1 - i)];
        append_bytes(be, fr_bytes);
    }
}
void EthereumHostTranscript::append_commitment(const char*, const uint8_t* x_le, const uint8_t* y_le, size_t fq_bytes,
                                               bool infinity) {
    uint8_t be[64];
    memset(be, 0, sizeof(be));
    if (!infinity)
        for (size_t i = 0; i < fq_bytes; ++i) be[i] = x_le[fq_bytes - 1 - i];
    append_bytes(be, fq_bytes);
    memset(be, 0, sizeof(be));
    if (!infinity) {
        for (size_t i = 0; i < fq_bytes; ++i) be[i] = y_le[fq_bytes - 1 - i];
    } else {
        be[fq_bytes - 1] = 1;  // GroupAffine::zero() has y = 1
    }
    append_bytes(be, fq_bytes);
}
void EthereumHostTranscript::challenge_scalar(const char*, size_t, uint8_t out_le[32]) {
    uint8_t data[1 + 64 + 4];
    data[0] = 2;
    memcpy(data + 1, state0_, 32);
    memcpy(data + 33, state1_, 32);
    data[65] = (uint8_t)(counter_ >> 24);
    data[66] = (uint8_t)(counter_ >> 16);
    data[67] = (uint8_t)(counter_ >> 8);
    data[68] = (uint8_t)counter_;
    ++counter_;
    uint8_t h[32];
    keccak256(data, sizeof(data), h);
    for (int i = 0; i < 32; ++i) out_le[i] = h[31 - i];
    out_le[31] &= 0x1f;
}

}  // namespace zkt
